#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tests/devtools/microbench/valu_rate 2>&1 | tee gpurun_out/ab3_valu_rate.txt
./tests/devtools/microbench/lds_rate 2>&1 | tee gpurun_out/ab3_lds_rate.txt
timeout 2400 python -m pytest tests/test_e2e_scripts_gpu.py -x -q 2>&1 | tail -60 | tee gpurun_out/ab3_e2e.txt
