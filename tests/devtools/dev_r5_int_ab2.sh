#!/bin/bash
# Round 5: A/B of build variants of the ray-centric pixel pass (libraries built with GOF_BUILD_TAG / GOF_EXTRA_FLAGS, selected with GOF_HIP_LIB):
#   gpurun --timeout 500 -- 'bash tests/devtools/dev_r5_int_ab2.sh shipped b256 noexit w6'
# First a small bit-exactness run of the shipped library under a short timeout (a kernel that hangs must not take the box with it).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5_int2; mkdir -p $O
PKG=gaussian-opacity-fields_amd
timeout 150 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "integrate_bit_exact_on_scene" > $O/pytest_small.txt 2>&1 || { tail -5 $O/pytest_small.txt; echo "SMALL TESTS FAILED OR TIMED OUT: not running the benchmarks"; exit 1; }
tail -1 $O/pytest_small.txt
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip.so; [ $v = shipped ] || lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip_$v.so
  echo "== $v"
  GOF_HIP_LIB=$lib timeout 200 python tests/devtools/dev_integrate_cache_bench.py 2> $O/$v.err | grep "call 0" | sed -e "s/'preprocess_fwd.*'integrate_pixels'/'integrate_pixels'/"
done > $O/ab.txt 2>&1
cat $O/ab.txt
