#!/bin/bash
# Round 5, GPU call A: (1) the parity report on HEAD's kernels in the shipped forward mode (tests/devtools/dev_parity_report.py ->
# gpurun_out/r05_parity_report.json), (2) a kernel trace of the full training iteration in the unchanged train.py composition.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/full_loop -- python $GRAFT_REPO_ROOT/tests/devtools/dev_full_loop_trace.py inline > $O/full_loop.txt 2> $O/full_loop.err ) || tail -3 $O/full_loop.err
cat $O/full_loop.txt
f=$(find $O/full_loop -name "*kernel_trace.csv" | head -1)
python tests/devtools/dev_trace_summary.py $f --marker adam_step --iters 10 > $O/full_loop_kernel_stats.md 2> $O/summary.err || tail -3 $O/summary.err
head -50 $O/full_loop_kernel_stats.md
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
timeout 900 python tests/devtools/dev_parity_report.py > $O/parity_report.txt 2> $O/parity_report.err || tail -5 $O/parity_report.err
tail -40 $O/parity_report.txt
