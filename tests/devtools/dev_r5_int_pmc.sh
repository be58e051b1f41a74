#!/bin/bash
# Round 5: SQ counters of the opacity-field query's kernels (one counter group per rocprofv3 run, never combined with sys / hip traces)
#   gpurun --timeout 600 -- 'bash tests/devtools/dev_r5_int_pmc.sh [s5m]'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5_int_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; ( cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag -- python $GRAFT_REPO_ROOT/tests/devtools/dev_pmc_integrate.py $SCENE > /tmp/pmc_$tag.log 2>&1 ) || tail -5 /tmp/pmc_$tag.log; }
SCENE=$1
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES
find $O -name "*agent_info.csv" -delete
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r5_int_pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("gof::", "")
        if not k.startswith("integrate"): continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in acc.items():
    print(k, {n: "%.4g" % v for n, v in sorted(c.items())})
    g = c.get
    if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        print("   VALU active / busy (per SIMD: x4 / CU-cycles):", g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES"), " inst-any active/wave-cycles:", g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES", 1))
    if g("SQ_INSTS_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print("   lane utilisation:", g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_INSTS_VALU")) , " wait_any/wave_cycles needs both passes; WAIT_INST_ANY/WAIT_ANY:", g("SQ_WAIT_INST_ANY", 0), g("SQ_WAIT_ANY", 0))
PY
