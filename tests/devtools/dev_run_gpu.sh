cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_mtets_gpu.py -m gpu -q -x --tb=short 2>&1 | cut -c1-300 | tail -6 ) 2>&1 | tail -8
