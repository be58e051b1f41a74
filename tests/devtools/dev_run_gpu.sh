cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -30
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench5.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench5.json'))
print(d["value"], d["ms_per_step"], d["full_loop"]["ms_per_iter"], {k:v["avg_ms"] for k,v in d["roofline"]["kernels"].items()})
PY
timeout 600 python tests/devtools/dev_integrate_cache_bench.py 2>&1 | grep -E "call [01]"
