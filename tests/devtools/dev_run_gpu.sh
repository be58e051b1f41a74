cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_train_epilogue_gpu.py -m gpu -q -x -k "end_to_end_training" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-300 | head
