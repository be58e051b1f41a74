cd $GRAFT_REPO_ROOT
timeout 300 python tests/devtools/dev_soak.py 600 2>&1 | tail -2
