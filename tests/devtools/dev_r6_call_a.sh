#!/bin/bash
# Round 6, GPU call A (1x MI355X): the round's new tests first (parity above 1M Gaussians, two streams / two threads, per-call modes),
# the default bench line, then the whole GPU suite WITHOUT -x and smoke.  Outputs under gpurun_out/r06a/.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06a
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x --tb=short -k "large_p or two_streams or per_call_modes" > $O/new_tests.txt 2>&1; tail -25 $O/new_tests.txt | cut -c1-1500
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
timeout 1700 python -m pytest tests -q -m gpu -rf --tb=short --durations=25 > $O/tests_full.txt 2>&1; tail -70 $O/tests_full.txt | cut -c1-1200
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.txt 2>&1; cat $O/smoke.txt
