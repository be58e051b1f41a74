#!/bin/bash
# The unchanged train.py at the headline resolution: 1600x1063 images rendered from 600k small ground-truth Gaussians, 400k initial
# points, 3000 iterations with densification; iterations/s of the real loop (tqdm), PSNR, Gaussian count; then extract_mesh.py.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak2
export PYTHONPATH=$PWD/tests/e2e_shims
S=/tmp/soak2_scene; M=/tmp/soak2_model
( time python tests/fixtures/make_blender_scene.py $S --views 40 --size 1600 1063 --gt 600000 --gt-scale 0.004 --init 400000 ) 2>&1 | grep real
( time python gaussian-opacity-fields_amd/launch/run_reference_script.py oracle/_ref/refpy/train.py -s $S -m $M --iterations 3000 --densify_from_iter 500 \
   --densification_interval 100 --opacity_reset_interval 1500 --densify_until_iter 2500 --distortion_from_iter 2000 --depth_normal_from_iter 2000 \
   --test_iterations 1 1000 3000 --save_iterations 3000 --eval ) > gpurun_out/soak2/train.log 2>&1
grep -a "Evaluating test\|real\|Training complete\|Number of points" gpurun_out/soak2/train.log | cut -c1-120
tr '\r' '\n' < gpurun_out/soak2/train.log | grep -a "Training progress" | awk 'NR%40==0' | cut -c1-140 | tail -8
python - <<'PY'
import sys; sys.path.insert(0,'tests/e2e_shims')
from plyfile import PlyData
print("gaussians in the saved model:", len(PlyData.read('/tmp/soak2_model/point_cloud/iteration_3000/point_cloud.ply')['vertex']))
PY
