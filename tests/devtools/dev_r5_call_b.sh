#!/bin/bash
# Round 5, GPU call B: (1) a parity subset on the shipped library and on the variant that sends every sort through the single-kernel
# passes, (2) the single-process A/B of the binning / backward-staging variants (tests/devtools/dev_r5_binning_ab.py),
# (3) the opacity-field query's cached-call bench on the shipped library and on that variant.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; rm -rf $O; mkdir -p $O
PKG=$GRAFT_REPO_ROOT/gaussian-opacity-fields_amd
K="forward_bit_exact or full_size_s1m or backward_blend_gradients or integrate_bit_exact or fused_forward or learnt_mask_pool or empty_and_culled"
( timeout 420 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "$K" 2>&1 | tail -5 ) > $O/tests_shipped.txt 2>&1
( GOF_HIP_LIB=$PKG/lib/libgof_hip_u16k_lb16.so timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "forward_bit_exact or full_size_s1m_against or full_size_s1m_clustered or integrate_bit_exact or integrate_full_size_s1m" 2>&1 | tail -5 ) > $O/tests_u16k.txt 2>&1
cat $O/tests_shipped.txt $O/tests_u16k.txt
timeout 420 python tests/devtools/dev_r5_binning_ab.py old:old shipped: lb1:lb1 lb16:lb16 u16k_lb8:u16k_lb8 u16k_lb16:u16k_lb16 u16k_lb32:u16k_lb32 \
   three_launch_scan::GOF_FUSED_SCAN=0 u16k_lb16_oshist:u16k_lb16:GOF_EMIT_HIST=0 bwstage1:bwstage1 shipped_again: > $O/ab.txt 2> $O/ab.err
tail -3 $O/ab.err; cat $O/ab.txt
for v in "" _u16k_lb16; do
  echo "== integrate, library '$v'"
  GOF_HIP_LIB=$PKG/lib/libgof_hip$v.so timeout 200 python tests/devtools/dev_integrate_cache_bench.py 2>&1 | tail -12
done > $O/integrate.txt 2>&1
cat $O/integrate.txt
