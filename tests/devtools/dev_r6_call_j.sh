#!/bin/bash
# Round 6, GPU call J (1x MI355X): the fixed deferred-loss GPU test, then the config-3-shaped trajectory (profiles/r06_trajectory_large.json's scene and
# arguments) for the launcher's defaults with the deferred loss and with GOF_EAGER_LOSS=1 on the same box -> gpurun_out/r06/trajectory_deferred.json
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06j
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_train_epilogue_gpu.py -q -m gpu -x --tb=short -k "deferred" > $O/new_tests.txt 2>&1; tail -5 $O/new_tests.txt | cut -c1-600
timeout 1500 python tests/devtools/dev_r6_trajectory.py --gt 400000 --gt-scale 0.007 --grad-threshold 0.00012 --runs product_default,product_default_eager_loss,product_default \
    --out $GRAFT_REPO_ROOT/gpurun_out/r06/trajectory_deferred.json > $O/trajectory.txt 2>&1; tail -12 $O/trajectory.txt | cut -c1-1600
