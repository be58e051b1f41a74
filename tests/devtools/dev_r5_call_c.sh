#!/bin/bash
# Round 5, GPU call C: parity subsets on the shipped library (histogram stride = the blocks that hold items) and with the binning chain on
# the library's high-priority stream (GOF_K1_SPLIT=2); then three A/B processes of at most three library instances each
# (tests/devtools/dev_r5_binning_ab.py): old / shipped / shipped with GOF_K1_SPLIT=2; tile sizes of the fused gather + scan and of the
# radix sort; os_hist's counting.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5c; rm -rf $O; mkdir -p $O
( timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "full_size_s1m_against_oracle or full_size_s1m_clustered or fused_forward or learnt_mask_pool or backward_blend_gradients" 2>&1 | tail -4 ) > $O/tests_shipped.txt 2>&1
( GOF_K1_SPLIT=2 timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "forward_bit_exact or full_size_s1m_against_oracle or fused_forward or learnt_mask_pool or backward_is_bit_reproducible" 2>&1 | tail -4 ) > $O/tests_split2.txt 2>&1
cat $O/tests_shipped.txt $O/tests_split2.txt
timeout 300 python tests/devtools/dev_r5_binning_ab.py old:old shipped: chain_high_prio::GOF_K1_SPLIT=2 > $O/ab_a.txt 2> $O/ab_a.err; tail -2 $O/ab_a.err
AB_SCENES=S1M,S1M-clustered,6M@1237x822 timeout 300 python tests/devtools/dev_r5_binning_ab.py gs4:gs4 gs8:gs8 rs512:rs512 > $O/ab_b.txt 2> $O/ab_b.err; tail -2 $O/ab_b.err
AB_SCENES=S1M,S1M-clustered timeout 200 python tests/devtools/dev_r5_binning_ab.py oshist_plain:oshplain chain_high_prio::GOF_K1_SPLIT=2 shipped: > $O/ab_c.txt 2> $O/ab_c.err; tail -2 $O/ab_c.err
cat $O/ab_a.txt $O/ab_b.txt $O/ab_c.txt | cut -c1-420
