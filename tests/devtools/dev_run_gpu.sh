cd $GRAFT_REPO_ROOT
timeout 600 python tests/devtools/dev_knn_time.py 2>&1 | tail -3
