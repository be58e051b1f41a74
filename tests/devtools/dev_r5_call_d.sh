#!/bin/bash
# Round 5, GPU call D: A/B of the emission leaving the tile sort's first histogram (GOF_EMIT_HIST0) and the per-Gaussian kernel clearing the
# depth sort's scratch (GOF_K1_ZERO) against the validated library (lib/libgof_hip_prev.so), two processes of three library instances.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; rm -rf $O; mkdir -p $O
( timeout 200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "capacity_above_the_count or fused_forward_matches or full_size_s1m_clustered" 2>&1 | tail -3 ) > $O/tests.txt 2>&1; cat $O/tests.txt
timeout 300 python tests/devtools/dev_r5_binning_ab.py prev:prev new: new_memset::GOF_K1_ZERO=0 > $O/ab_a.txt 2> $O/ab_a.err
timeout 300 python tests/devtools/dev_r5_binning_ab.py new: new_rs_hist::GOF_EMIT_HIST0=0 prev:prev > $O/ab_b.txt 2> $O/ab_b.err
cat $O/ab_a.txt $O/ab_b.txt | cut -c1-400
