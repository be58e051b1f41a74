cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py tests/test_host_api.py -m gpu -q -k "not integrate and not config5" 2>&1 | tail -1
timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -1 | cut -c1-330
