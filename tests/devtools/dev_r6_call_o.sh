#!/bin/bash
# Round 6, GPU call O: host-side trims of the training iteration (the raw current-stream query in the bindings, the loss terms read back without a launch):
# the deferred-loss and epilogue GPU tests, the stream tests, then the config-3-shaped trajectory with the launcher's defaults (it/s against 251.3 before).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06o; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_train_epilogue_gpu.py tests/test_parity_gpu.py -q -m gpu -x --tb=short -k "deferred or end_to_end or two_streams or per_call_modes or fused_adam" > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-600
timeout 1200 python tests/devtools/dev_r6_trajectory.py --gt 400000 --gt-scale 0.007 --grad-threshold 0.00012 --runs product_default --out $O/trajectory.json > $O/trajectory.txt 2>&1; tail -3 $O/trajectory.txt | cut -c1-400
