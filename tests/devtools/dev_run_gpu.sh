cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py -m gpu -q -k "integrate" 2>&1 | tail -1
timeout 600 python tests/devtools/dev_integrate_cache_bench.py 2>&1 | grep -v "amdgpu.ids" | tail -12 > gpurun_out/integrate_cache_bench.log; grep -E "call [01]" gpurun_out/integrate_cache_bench.log | cut -c1-330
