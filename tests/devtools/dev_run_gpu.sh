cd $GRAFT_REPO_ROOT
for i in 1 2; do
GOF_FUSED_FORWARD=0 timeout 300 python tests/devtools/dev_event_overhead.py 2>&1 | tail -1
GOF_FUSED_FORWARD=1 timeout 300 python tests/devtools/dev_event_overhead.py 2>&1 | tail -1
done
