#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$PWD/gaussian-opacity-fields_amd/lib
for v in _stats _statsw; do echo "== bwstats $v"; GOF_HIP_LIB=$L/libgof_hip$v.so timeout 300 python tests/devtools/dev_bwstats.py 2>&1 | grep -v amdgpu.ids | tail -3; done | tee gpurun_out/ab2_bwstats.txt
for v in "" _wavewalk _c128 _c256; do
  echo "== variant '$v'"
  GOF_HIP_LIB=$L/libgof_hip$v.so timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -2
done 2>&1 | tee gpurun_out/ab2_time.txt
timeout 2400 python -m pytest tests/test_e2e_scripts_gpu.py -x -q 2>&1 | tail -60 | tee gpurun_out/ab2_e2e.txt
