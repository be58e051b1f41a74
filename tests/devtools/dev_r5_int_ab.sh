#!/bin/bash
# Round 5: A/B of the opacity-field query's pixel pass -- ray-centric (shipped: every distinct sub-ray of a tile once, integrate_rays) against
# pixel-centric (rounds 1-4, GOF_INT_PIXELS=1) -- with tests/devtools/dev_integrate_cache_bench.py (S1M + 9M points; config-5 shape: 5M
# Gaussians + 45M points), then the integrate / mesh-extraction parity tests on the shipped form.
#   gpurun --timeout 700 -- 'bash tests/devtools/dev_r5_int_ab.sh'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5_int; mkdir -p $O
for v in 0 1; do
  echo "== GOF_INT_PIXELS=$v"
  GOF_INT_PIXELS=$v timeout 250 python tests/devtools/dev_integrate_cache_bench.py 2> $O/mode$v.err
done > $O/ab.txt 2>&1
timeout 400 python -m pytest tests/test_parity_gpu.py tests/test_mesh_extraction_gpu.py -q -x -m gpu -k "integrate or mesh" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
cat $O/ab.txt
