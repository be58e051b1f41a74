cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "integrate_full_size_s1m" --durations=3 2>&1 | grep -E "passed|failed|^E  |s call" | cut -c1-300 | head
