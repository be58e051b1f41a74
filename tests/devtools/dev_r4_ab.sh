#!/bin/bash
# Round 4, A/B of the blend kernels' variants in ONE GPU call (variant libraries are built on the CPU box by
# `bash tests/devtools/dev_r4_ab.sh build` and travel with the snapshot; selected with GOF_HIP_LIB):
#   shipped            forward: fp32 value path (gexpf) + redo list; backward: wave reduction through the LDS (GOF_BW_REDUCE=1)
#   shipped, exact     the same library in the forward's verification mode (GOF_FW_EXACT=1) = round 3's forward arithmetic
#   hwexp              -DGOF_FW_HWEXP       forward exponential by v_exp_f32
#   bw0                -DGOF_BW_REDUCE=0    round 3's register reduction (v_permlane swaps)
#   bw2                -DGOF_BW_REDUCE=2    LDS reduction in two halves (5 workgroups per CU)
#   tight              -DGOF_TIGHT_RECTS    tile rectangle intersected with the footprint box
#   gpurun --timeout 900 -- 'bash tests/devtools/dev_r4_ab.sh'
PKG=gaussian-opacity-fields_amd
declare -A FLAGS=( [hwexp]="-DGOF_FW_HWEXP" [bw0]="-DGOF_BW_REDUCE=0" [bw2]="-DGOF_BW_REDUCE=2" [tight]="-DGOF_TIGHT_RECTS" [hwexp_tight]="-DGOF_FW_HWEXP -DGOF_TIGHT_RECTS" )
if [ "$1" = build ]; then
  cd "$(dirname "$0")/../.."
  for v in "${!FLAGS[@]}"; do GOF_BUILD_TAG=$v GOF_EXTRA_FLAGS="${FLAGS[$v]}" python $PKG/build.py & done; wait
  python $PKG/build.py; ls -la $PKG/lib; exit 0
fi
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_ab; mkdir -p $O
SCENES=${SCENES:-"s1m clustered"}
for scene in $SCENES; do
  for v in shipped exact hwexp bw0 bw2 tight hwexp_tight; do
    lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip.so; ex=0
    case $v in shipped) ;; exact) ex=1 ;; *) lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip_$v.so ;; esac
    [ -f $lib ] || { echo "== $scene $v: $lib missing"; continue; }
    echo "== $scene $v"
    GOF_HIP_LIB=$lib GOF_FW_EXACT=$ex timeout 200 python tests/devtools/dev_time.py $scene 2> $O/${scene}_$v.err | tail -3
  done
done 2>&1 | tee $O/ab_time.txt
