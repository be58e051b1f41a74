#!/bin/bash
# Round 4, late: A/B of the opacity-field query's point pass -- the point's ray stored once per call by gather_sorted_points (shipped)
# against formed from the pixel position in every staged batch (libgof_hip_intold.so = the previous commit) -- with
# tests/devtools/dev_integrate_cache_bench.py (S1M + 9M points, config-5 shape: 5M Gaussians + 45M points), then the integrate parity tests.
#   gpurun --timeout 600 -- 'bash tests/devtools/dev_r4_int_ab.sh'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_int; mkdir -p $O
PKG=gaussian-opacity-fields_amd
for v in shipped intold; do
  lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip.so; [ $v = shipped ] || lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip_$v.so
  echo "== $v"
  GOF_HIP_LIB=$lib timeout 250 python tests/devtools/dev_integrate_cache_bench.py 2> $O/$v.err
done > $O/ab.txt 2>&1
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_mesh_extraction_gpu.py -q -x -m gpu -k "integrate or mesh" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
cat $O/ab.txt
