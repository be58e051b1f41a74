#!/bin/bash
# Round 6, GPU call P: the deferred loss with a carried tensor term (train.py's decoupled-appearance L1): the end-to-end run of the unchanged script with
# --use_decoupled_appearance, the other end-to-end training tests and the deferred-loss GPU test.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06p; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_e2e_scripts_gpu.py tests/test_train_epilogue_gpu.py -q -m gpu --tb=short --durations=6 -k "appearance or train_py or train_alike or deferred" > $O/tests.txt 2>&1; tail -30 $O/tests.txt | cut -c1-1500
