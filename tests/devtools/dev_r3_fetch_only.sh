cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof3b
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
NOPYTEST=1 LIBS=product SCENES="s1m s1m_clustered" bash tests/devtools/dev_r3_sched.sh > /dev/null 2>&1
tail -2 gpurun_out/r3_sched.jsonl | cut -c1-600
( cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/fetch -- python $GRAFT_REPO_ROOT/tests/devtools/dev_pmc.py > /tmp/pmc_fetch.log 2>&1 ) || tail -5 /tmp/pmc_fetch.log
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof3b/pmc/fetch/**/*counter_collection.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='FETCH_SIZE': d[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k,v in sorted(d.items(),key=lambda kv:-sum(kv[1])/len(kv[1]))[:8]: print(k, round(2*sum(v)/len(v)*1024/1e6,1),'MB (x2)')
PY
