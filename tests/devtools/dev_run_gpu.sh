cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "backward_fuzz" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -20
