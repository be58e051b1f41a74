cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_knn.py -m gpu -q -k "forward_bit or properties_s1m or empty or knn or matches" 2>&1 | tail -1
timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -1 | cut -c1-330
