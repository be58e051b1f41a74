#!/bin/bash
# Round 4, A/B of the blend kernels' variants in ONE GPU call (variant libraries are built on the CPU box by
# `bash tests/devtools/dev_r4_ab.sh build` and travel with the snapshot; selected with GOF_HIP_LIB):
#   shipped            the library as built by build.py (forward: division-free default mode; backward: GOF_BW_REDUCE as set in blend_backward.hip)
#   exact / tight      the same library with GOF_FW_EXACT=1 (the forward's verification mode = round 3's arithmetic) / GOF_TIGHT_RECTS=1
#   bw0 .. bw3z        -DGOF_BW_REDUCE=0|1|2|3 (+ -DGOF_BW_ZSTORE): the backward's wave reduction (registers | LDS | LDS in halves | LDS pipelined)
#   fw5                -DGOF_FW_WAVES=5
#   gpurun --timeout 900 -- 'bash tests/devtools/dev_r4_ab.sh'
PKG=gaussian-opacity-fields_amd
declare -A FLAGS=( [bw0]="-DGOF_BW_REDUCE=0" )     # (call 2 also timed -DGOF_BW_REDUCE=1|3 (+ -DGOF_BW_ZSTORE) and -DGOF_FW_WAVES=5: profiles/r04_ab_call2_*.txt; those variants were removed)
if [ "$1" = build ]; then
  cd "$(dirname "$0")/../.."
  for v in "${!FLAGS[@]}"; do GOF_BUILD_TAG=$v GOF_EXTRA_FLAGS="${FLAGS[$v]}" python $PKG/build.py & done; wait
  python $PKG/build.py; ls -la $PKG/lib; exit 0
fi
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_ab; mkdir -p $O
SCENES=${SCENES:-"s1m clustered"}
for scene in $SCENES; do
  for v in ${VARIANTS:-shipped exact tight bw0}; do
    lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip.so; ex=0; tr=0
    case $v in shipped) ;; exact) ex=1 ;; tight) tr=1 ;; *) lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip_$v.so ;; esac
    [ -f $lib ] || { echo "== $scene $v: $lib missing"; continue; }
    echo "== $scene $v"
    GOF_HIP_LIB=$lib GOF_FW_EXACT=$ex GOF_TIGHT_RECTS=$tr timeout 200 python tests/devtools/dev_time.py $scene 2> $O/${scene}_$v.err | tail -3
  done
done 2>&1 | tee $O/ab_time.txt
