cd $GRAFT_REPO_ROOT
timeout 600 python tests/devtools/dev_skew.py 2>&1 | tail -2
