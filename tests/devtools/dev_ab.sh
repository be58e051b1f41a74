#!/bin/bash
# A/B of library variants (GOF_BUILD_TAG builds in lib/): dev_ab.sh "<tag> <tag> ..." ('' = the shipped build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" $VARIANTS; do
  lib=gaussian-opacity-fields_amd/lib/libgof_hip${v:+_$v}.so
  echo "== variant '$v'"
  GOF_HIP_LIB=$PWD/$lib python tests/devtools/dev_time.py 2>/dev/null | tail -2
done 2>&1 | tee gpurun_out/ab_time.txt
