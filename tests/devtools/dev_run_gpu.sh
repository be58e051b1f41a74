cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "fused_forward" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-300 | head
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout 300 python tests/devtools/dev_event_overhead.py 2>&1 | tail -2
