#!/bin/bash
# Round-2 A/B session: parity suite on the product library, then per-kernel timing of the library variants in lib/ at S1M.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/ab_pytest.txt
cat gpurun_out/ab_pytest.txt
for v in "" _r1 _b256 $EXTRA_VARIANTS; do
  echo "== variant '$v'"
  GOF_HIP_LIB=$PWD/gaussian-opacity-fields_amd/lib/libgof_hip$v.so timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -2
done 2>&1 | tee gpurun_out/ab_time.txt
GOF_HIP_LIB=$PWD/gaussian-opacity-fields_amd/lib/libgof_hip_stats.so timeout 300 python tests/devtools/dev_fwstats.py 2>&1 | tail -5 | tee gpurun_out/ab_fwstats.txt
