#!/bin/bash
# Round 6, GPU call B: A/B of (i) a BOUNDED grid for stage 2 of the per-Gaussian kernel beside the binning chain (GOF_K1_HEAVY_WGS_PER_CU = 1 / 2 / 4
# workgroups per CU; GOF_K1_SPLIT=0: no fork at all), (ii) more waves per tile in the single-kernel radix passes and the fused gather + scan
# (GOF_OS_WAVES, GOF_GS_WAVES).  tests/devtools/dev_r5_binning_ab.py: at most three library instances per process.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06b
rm -rf $O; mkdir -p $O
export AB_SCENES=${AB_SCENES:-S1M,S1M-clustered,6M@1237x822}
timeout 400 python tests/devtools/dev_r5_binning_ab.py shipped: h1:h1 h2:h2 > $O/ab_a.txt 2> $O/ab_a.err; tail -2 $O/ab_a.err
timeout 400 python tests/devtools/dev_r5_binning_ab.py h4:h4 nosplit:nosplit shipped: > $O/ab_b.txt 2> $O/ab_b.err; tail -2 $O/ab_b.err
timeout 400 python tests/devtools/dev_r5_binning_ab.py osw16:osw16 shipped: w16:w16 > $O/ab_c.txt 2> $O/ab_c.err; tail -2 $O/ab_c.err
timeout 400 python tests/devtools/dev_r5_binning_ab.py shipped: osw8:osw8 h2w16:h2w16 > $O/ab_d.txt 2> $O/ab_d.err; tail -2 $O/ab_d.err
cat $O/ab_a.txt $O/ab_b.txt $O/ab_c.txt $O/ab_d.txt | cut -c1-900
