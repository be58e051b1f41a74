#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py -q -x -k "integrate or uint16 or parameter_gradients or cull_scan" 2>&1 | grep -v "ERROR: Maximal" | tail -30 | tee gpurun_out/int_pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/int_bench.txt
import sys, json
sys.path.insert(0, 'gaussian-opacity-fields_amd'); sys.path.insert(0, 'tests')
import torch, bench
print(json.dumps(bench.integrate_leg(torch.device('cuda', 0))))
PY
python tests/devtools/dev_integrate_cache_bench.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/int_cache_bench.txt
