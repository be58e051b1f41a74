cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "no_visible" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-250 | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
