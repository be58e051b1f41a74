#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tests/devtools/dev_densify_check.py 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-700 | tee gpurun_out/b10_densify.txt
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_e2e_scripts_gpu.py tests/test_train_epilogue_gpu.py -q -x -k "integrate or uint16 or e2e or scripts or densify or train_py or extract or rank" 2>&1 | grep -v "ERROR: Maximal" | tail -30 | tee gpurun_out/b10_pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b10_bench.txt
import sys, json
sys.path.insert(0, 'gaussian-opacity-fields_amd'); sys.path.insert(0, 'tests')
import torch, bench
print(json.dumps(bench.integrate_leg(torch.device('cuda', 0))))
PY
python tests/devtools/dev_integrate_cache_bench.py 2>&1 | grep -v amdgpu.ids | grep "call 0\|call 1" | tee gpurun_out/b10_cache_bench.txt
