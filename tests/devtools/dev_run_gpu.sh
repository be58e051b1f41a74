cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_train_epilogue_gpu.py -m gpu -q -k "densification or 3d_filter" 2>&1 | tail -1
