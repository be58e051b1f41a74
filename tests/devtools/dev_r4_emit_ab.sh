#!/bin/bash
# Round 4, late: A/B of emit_instances -- output slots divided among the waves (shipped; -DGOF_EMIT_SLOTS=512 / 2048 variants) against
# the round-1..4 kernel in which a wave owned 64 depth-sorted Gaussians (libgof_hip_emitold.so, built from the previous commit) -- on
# S1M and S1M-clustered, then the binning-related parity tests on the shipped library.
#   bash tests/devtools/dev_r4_ab.sh build-style variants are built on the CPU box (see the header of dev_r4_ab.sh); then
#   gpurun --timeout 600 -- 'bash tests/devtools/dev_r4_emit_ab.sh'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_emit; mkdir -p $O
VARIANTS="shipped emitold emit512 emit2048" SCENES="s1m clustered" bash tests/devtools/dev_r4_ab.sh > $O/ab.txt 2>&1
cp gpurun_out/r4_ab/ab_time.txt $O/ab_time.txt
timeout 400 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "forward_bit_exact or full_size_s1m_clustered or full_size_s1m_against or fused_forward_matches or empty_and_culled or learnt_mask_pool or integrate_bit_exact_on_scene" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
cat $O/ab_time.txt
