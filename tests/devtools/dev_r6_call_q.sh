#!/bin/bash
# Round 6, GPU call Q: the config-3-shaped trajectory at a LARGER scale -- 64 + 8 views of 1200x1200 from 1.2M ground-truth Gaussians of 0.005 units, 200k initial
# points, densify_grad_threshold 1e-4 -- product / the reference's own kernels (twice) / the launcher's defaults, 7 000 iterations of the unchanged train.py.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06q; rm -rf $O; mkdir -p $O
timeout 3000 python tests/devtools/dev_r6_trajectory.py --size 1200 1200 --gt 1200000 --gt-scale 0.005 --init 200000 --grad-threshold 0.0001 \
    --runs product_default,product,reference,reference2 --out $O/trajectory_xl.json > $O/trajectory.txt 2>&1; tail -8 $O/trajectory.txt | cut -c1-900
