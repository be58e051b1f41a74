#!/bin/bash
# Round 6, GPU call L: integrate_points with the packed prelude (the pair layout of blend_forward's phase 2: normal, AA, BB in 12 instructions instead of 21)
# against the scalar form (lib/libgof_hip_ipold.so = HEAD's integrate.hip): the query's bit-exactness tests under the new default first, then the config-5-shaped
# query timed under both libraries, alternating, three times each (tests/devtools/dev_integrate_cache_bench.py: call 0 = first call of a view, later calls cached).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06l; rm -rf $O; mkdir -p $O
PKG=gaussian-opacity-fields_amd
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "integrate" > $O/pytest.txt 2>&1 || { tail -15 $O/pytest.txt; echo "INTEGRATE TESTS FAILED"; exit 1; }
tail -1 $O/pytest.txt
for rep in 1 2 3; do
for v in shipped ipold; do
  lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip.so; [ $v = shipped ] || lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip_$v.so
  echo "== $v ($rep)"
  GOF_HIP_LIB=$lib timeout 200 python tests/devtools/dev_integrate_cache_bench.py 2> $O/$v.err | cut -c1-600
done; done > $O/ab.txt 2>&1
cat $O/ab.txt
