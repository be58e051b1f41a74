cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "fuzz or integrate_scenes or determin or s10k or config" --tb=short 2>&1 | cut -c1-300 | tail -4
cd tests/devtools && python dev_mtets_time.py 2>&1 | tail -3
