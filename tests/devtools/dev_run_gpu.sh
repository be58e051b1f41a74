cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py -m gpu -q -k "backward or gradient or s1m_against or determin" 2>&1 | tail -1
timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -1 | cut -c1-330
