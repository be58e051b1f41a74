#!/bin/bash
# PMC passes over the two blend kernels at S1M (tests/devtools/dev_pmc.py = 3 forward + backward iterations); one counter group per run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run() {  # $1 = tag, rest = counters
  tag=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag -- python $GRAFT_REPO_ROOT/tests/devtools/dev_pmc.py > /tmp/pmc_$tag.log 2>&1 ) || tail -5 /tmp/pmc_$tag.log
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES
run sq4 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
python - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc")
for tag in sorted(os.listdir(root)):
    fs = glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print(tag, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "blend_forward" in k or "blend_backward" in k:
            acc["fwd" if "forward" in k else "bwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kern, d in acc.items():
        print(tag, kern, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
PY
