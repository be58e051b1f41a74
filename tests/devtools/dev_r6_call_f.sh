#!/bin/bash
# Round 6, GPU call F: (i) the scans of the radix sorts' [digit][block] histograms in tiles of 1024 / 2048 / 4096 words (GOF_SCAN_SMALL_ITEMS 4 / 8 / 16),
# (ii) blend_forward's phase 2 with one divergent region per trip (GOF_FW_FLAT=1); interleaved (tests/devtools/dev_r6_ab.py).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06f
rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py -q -m gpu -x --tb=short -k "forward_bit_exact or large_p or fused_forward_at_full_size" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-600
timeout 600 python tests/devtools/dev_r6_ab.py scan4: scan8:scan8 scan16:scan16 > $O/ab_a.txt 2> $O/ab_a.err; tail -2 $O/ab_a.err
timeout 600 python tests/devtools/dev_r6_ab.py flat:flat shipped: > $O/ab_b.txt 2> $O/ab_b.err; tail -2 $O/ab_b.err
cat $O/ab_a.txt $O/ab_b.txt | cut -c1-620
