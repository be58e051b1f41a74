cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "share_one" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-250 | head
