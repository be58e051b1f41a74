cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-250 | head -30
timeout 600 python tests/dev_integrate_cache_bench.py 2>&1 | grep -v "amdgpu.ids" | tail -12 > gpurun_out/integrate_cache_bench.log; cat gpurun_out/integrate_cache_bench.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
