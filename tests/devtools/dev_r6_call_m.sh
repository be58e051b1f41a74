#!/bin/bash
# Round 6, GPU call M: the fp32 prelude of a (pixel, entry) pair (normal, AA, BB) with PLAIN instead of packed instructions in blend_forward's phase 2
# (GOF_FW_SCALAR_PRELUDE=1), blend_backward (GOF_BW_SCALAR_PRELUDE=1) and both, same LDS layout and same bits -- after the packed form measured 6.5 %
# slower in integrate_points (profiles/r06_ab_call8_*.txt).  Forward / backward parity tests under the both-scalar build first, then interleaved timing.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06m; rm -rf $O; mkdir -p $O
GOF_HIP_LIB=$GRAFT_REPO_ROOT/gaussian-opacity-fields_amd/lib/libgof_hip_bothsc.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x --tb=short -k "forward_bit_exact or backward or full_size_s1m" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-600
timeout 900 python tests/devtools/dev_r6_ab.py shipped: fwsc:fwsc bwsc:bwsc both:bothsc > $O/ab.txt 2> $O/ab.err; tail -2 $O/ab.err
cat $O/ab.txt | cut -c1-900
