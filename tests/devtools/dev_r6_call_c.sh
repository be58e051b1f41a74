#!/bin/bash
# Round 6, GPU call C: the stage-2 grid bound again, interleaved (tests/devtools/dev_r6_ab.py): shipped / 1 / 2 workgroups per CU / no fork
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06c
rm -rf $O; mkdir -p $O
timeout 600 python tests/devtools/dev_r6_ab.py shipped: h1:h1 nosplit:nosplit > $O/ab_a.txt 2> $O/ab_a.err; tail -2 $O/ab_a.err
timeout 600 python tests/devtools/dev_r6_ab.py h2:h2 shipped: h1:h1 > $O/ab_b.txt 2> $O/ab_b.err; tail -2 $O/ab_b.err
cat $O/ab_a.txt $O/ab_b.txt | cut -c1-700
