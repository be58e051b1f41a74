#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "backward or gradient or autograd or s1m" 2>&1 | tail -5 | tee gpurun_out/b11_pytest.txt
for v in "" _r1; do echo "== variant '$v'"; GOF_HIP_LIB=$PWD/gaussian-opacity-fields_amd/lib/libgof_hip$v.so timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -2; done | tee gpurun_out/b11_time.txt
