cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench10.json 2> gpurun_out/bench10.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof10 -o r10 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-loop > $R/gpurun_out/prof10.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc10_$c -o p -- python $R/tests/devtools/dev_pmc.py > $R/gpurun_out/pmc10_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc10_valu -o p -- python $R/tests/devtools/dev_pmc.py > $R/gpurun_out/pmc10_valu.log 2>&1
cd $R; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench10.json'))
print(d["value"], d["ms_per_step"], d["full_loop"]["ms_per_iter"], {k:v["avg_ms"] for k,v in d["roofline"]["kernels"].items()})
PY
timeout 600 python tests/devtools/dev_integrate_cache_bench.py 2>&1 | grep -v "amdgpu.ids" | tail -12 > gpurun_out/integrate_cache_bench.log
