cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) 2>&1 | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py > gpurun_out/bench_final.json ) 2>&1 | tail -3
