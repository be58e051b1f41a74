cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r4d; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -q --durations=8) > $O/gpu_suite.txt 2>&1
tail -4 $O/gpu_suite.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
