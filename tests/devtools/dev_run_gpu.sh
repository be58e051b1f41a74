cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "forward_bit or properties_s1m or fused" 2>&1 | tail -1
timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -1 | cut -c1-330
