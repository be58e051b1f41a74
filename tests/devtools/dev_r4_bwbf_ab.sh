#!/bin/bash
# Round 4, late: A/B of (a) gexpf without its never-firing upper clamp in the forward / the pixel pass (shipped vs libgof_hip_prev.so =
# the previous commit) and (b) the backward's gradient block run by every lane with zero stand-ins instead of cleared registers and a
# divergent branch (-DGOF_BW_BRANCHFREE=1: libgof_hip_bwbf.so); then the backward's tolerance / reproducibility tests ON the variant.
#   gpurun --timeout 600 -- 'bash tests/devtools/dev_r4_bwbf_ab.sh'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_bwbf; mkdir -p $O
VARIANTS="shipped prev bwbf shipped prev bwbf" SCENES="s1m" bash tests/devtools/dev_r4_ab.sh > $O/ab_s1m.txt 2>&1
cp gpurun_out/r4_ab/ab_time.txt $O/ab_time_s1m.txt
VARIANTS="shipped prev bwbf" SCENES="clustered" bash tests/devtools/dev_r4_ab.sh > $O/ab_cl.txt 2>&1
cp gpurun_out/r4_ab/ab_time.txt $O/ab_time_clustered.txt
GOF_HIP_LIB=$GRAFT_REPO_ROOT/gaussian-opacity-fields_amd/lib/libgof_hip_bwbf.so timeout 300 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "backward" > $O/pytest_bwbf.txt 2>&1
tail -4 $O/pytest_bwbf.txt
grep -h "==\|blend_forward" $O/ab_time_s1m.txt $O/ab_time_clustered.txt | sed -e "s/'preprocess_fwd.*'blend_forward'/'blend_forward'/" -e "s/'order_tiles_bw'.*'blend_backward'/'blend_backward'/" -e "s/, 'gather.*//"
