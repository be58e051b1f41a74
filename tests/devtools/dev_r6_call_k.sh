#!/bin/bash
# Round 6, GPU call K: the backward's two per-Gaussian launches as a pipeline of chunks (GOF_BW_CHUNKS = 2 / 4 / 8: gather_tile_partials of chunk c + 1
# on the caller's stream beside preprocess_bwd of chunk c on the library's second stream) against the two launches over all Gaussians; the backward
# parity tests under the 4-chunk build first (same bits expected), then interleaved timing (tests/devtools/dev_r6_ab.py).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06k
rm -rf $O; mkdir -p $O
GOF_HIP_LIB=$GRAFT_REPO_ROOT/gaussian-opacity-fields_amd/lib/libgof_hip_bwc4.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x --tb=short -k "backward or large_p or full_size_s1m" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-600
timeout 900 python tests/devtools/dev_r6_ab.py shipped: chunks2:bwc2 chunks4:bwc4 chunks8:bwc8 > $O/ab.txt 2> $O/ab.err; tail -2 $O/ab.err
cat $O/ab.txt | cut -c1-700
