cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mtets_gpu.py -m gpu -q -x --tb=short 2>&1 | cut -c1-300 | tail -5
(cd tests/devtools && python dev_mtets_time.py 2>&1 | tail -3)
