#!/bin/bash
# Round 6, GPU call I (1x MI355X): the launcher's deferred loss (train_epilogue/deferred.py) -- its GPU test, the end-to-end runs of the unchanged
# train.py (every iteration one fused loss call), the full-iteration bench legs and a kernel trace of what the launcher runs by default.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06i
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_train_epilogue_gpu.py -q -m gpu -x --tb=short -k "deferred or end_to_end or equals_the_composition" > $O/new_tests.txt 2>&1; tail -15 $O/new_tests.txt | cut -c1-1500
timeout 1500 python -m pytest tests/test_e2e_scripts_gpu.py -q -m gpu --tb=short -k "train_py or train_alike or data_parallel" > $O/e2e.txt 2>&1; tail -25 $O/e2e.txt | cut -c1-1500
timeout 900 python bench.py --no-cpu-baseline --no-integrate --no-clustered --no-views --no-reference --no-kernel-size-leg > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06i/bench.json"))
fl = d["full_loop"]
print(d["value"], d["ms_per_step"], {k: fl[k] for k in ("ms_per_iter", "iters_per_s")})
for k in ("launcher_default", "launcher_eager_loss", "one_call_loss", "one_call_loss_split_sh"):
    print(k, {a: b for a, b in fl[k].items() if a != "what"})
for k, v in d.get("large_p", {}).items():
    print(k, v.get("ms_per_step"), v.get("full_loop"))
PY
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/full_loop -- python $GRAFT_REPO_ROOT/tests/devtools/dev_full_loop_trace.py launcher > $O/full_loop.txt 2> $O/full_loop.err ) || tail -3 $O/full_loop.err
cat $O/full_loop.txt
f=$(find $O/full_loop -name "*kernel_trace.csv" | head -1)
python tests/devtools/dev_trace_summary.py $f --marker adam_kernel --iters 10 > $O/full_loop_kernel_stats.md 2> $O/summary.err || tail -3 $O/summary.err
head -60 $O/full_loop_kernel_stats.md | cut -c1-200
