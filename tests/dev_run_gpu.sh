cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_epilogue_gpu.py -m gpu -q 2>&1 | tail -5 > gpurun_out/ep_test.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench4.json 2> gpurun_out/bench4.err
tail -5 gpurun_out/ep_test.log; tail -3 gpurun_out/bench4.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench4.json'))
print(d["value"], d["ms_per_step"], d.get("full_loop"))
PY
