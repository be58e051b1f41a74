cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_epilogue_gpu.py -m gpu -q -x -k "end_to_end" --tb=short 2>&1 | cut -c1-400 | tail -8
