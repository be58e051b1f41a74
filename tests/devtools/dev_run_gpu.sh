cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "config5" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-300 | head
