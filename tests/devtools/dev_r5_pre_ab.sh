#!/bin/bash
# Round 5: A/B of build variants of the per-Gaussian stages (GOF_BUILD_TAG libraries selected with GOF_HIP_LIB)
#   gpurun --timeout 600 -- 'bash tests/devtools/dev_r5_pre_ab.sh shipped pre5 pre6 tiled tiled5'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5_pre; mkdir -p $O
PKG=gaussian-opacity-fields_amd
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip.so; [ $v = shipped ] || lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip_$v.so
  echo "== $v"
  GOF_HIP_LIB=$lib timeout 200 python tests/devtools/dev_pre_time.py 2> $O/$v.err
done > $O/ab.txt 2>&1
cat $O/ab.txt
