#!/bin/bash
# Round 4, late: blend_backward's LDS -- slab rows padded by one word (shipped) vs unpadded (libgof_hip_pad0.so), and 32 staged entries
# per batch instead of 64 (libgof_hip_b32.so: 20 KB of LDS and 79 VGPRs = 6 waves per SIMD instead of 5)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_bwlds; mkdir -p $O
VARIANTS="shipped pad0 b32 shipped pad0 b32" SCENES="s1m" bash tests/devtools/dev_r4_ab.sh > $O/ab_s1m.txt 2>&1; cp gpurun_out/r4_ab/ab_time.txt $O/ab_time_s1m.txt
VARIANTS="shipped pad0 b32" SCENES="clustered" bash tests/devtools/dev_r4_ab.sh > $O/ab_cl.txt 2>&1; cp gpurun_out/r4_ab/ab_time.txt $O/ab_time_clustered.txt
for v in "" _b32; do
GOF_HIP_LIB=$GRAFT_REPO_ROOT/gaussian-opacity-fields_amd/lib/libgof_hip$v.so timeout 300 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "backward" > $O/pytest$v.txt 2>&1; tail -2 $O/pytest$v.txt
done
grep -h "==\|blend_forward" $O/ab_time_s1m.txt $O/ab_time_clustered.txt | sed -e "s/'preprocess_fwd.*'blend_forward'/'blend_forward'/" -e "s/'order_tiles_bw'.*'blend_backward'/'blend_backward'/" -e "s/, 'preprocess_bwd.*//"
