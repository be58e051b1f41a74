cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "forward or s1m" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head
timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -1
