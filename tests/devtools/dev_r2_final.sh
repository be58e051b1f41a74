#!/bin/bash
# final validation: whole GPU suite, smoke, N=2 shared-GPU bench (spawn path + exchange report), default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "ERROR: Maximal" | tail -15 | tee gpurun_out/final/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/final/smoke.txt
GOF_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/final/bench_n2_shared.json 2> gpurun_out/final/bench_n2.err; tail -2 gpurun_out/final/bench_n2.err; cut -c1-600 gpurun_out/final/bench_n2_shared.json
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -2 gpurun_out/final/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['valu_issue_frac'], d['roofline']['traffic_source'])
PY
