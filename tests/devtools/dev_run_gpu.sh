cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
timeout 120 python tests/devtools/dev_time.py 2>&1 | tail -1 | cut -c1-330
