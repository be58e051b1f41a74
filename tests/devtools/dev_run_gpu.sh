cd $GRAFT_REPO_ROOT
timeout 600 python tests/devtools/dev_mtets_time.py 2>&1 | tail -4
