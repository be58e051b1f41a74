#!/bin/bash
# A longer end-to-end run of the unchanged train.py at a more realistic size: 800x600 images rendered from 40k ground-truth
# Gaussians, 100k initial points, 4000 iterations with densification -- iterations/s of the real loop, PSNR, Gaussian count.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
export PYTHONPATH=$PWD/tests/e2e_shims
S=/tmp/soak_scene; M=/tmp/soak_model
python tests/fixtures/make_blender_scene.py $S --views 60 --size 800 600 --gt 40000 --gt-scale 0.012 --init 100000 > /dev/null 2>&1
( time python gaussian-opacity-fields_amd/launch/run_reference_script.py oracle/_ref/refpy/train.py -s $S -m $M --iterations 4000 --densify_from_iter 300 \
   --densification_interval 100 --opacity_reset_interval 1500 --densify_until_iter 3000 --distortion_from_iter 2000 --depth_normal_from_iter 2000 \
   --test_iterations 1 1000 2000 4000 --save_iterations 4000 --eval ) > gpurun_out/soak/train.log 2>&1
grep -a "Evaluating\|real\|Training complete" gpurun_out/soak/train.log | cut -c1-120
tr '\r' '\n' < gpurun_out/soak/train.log | grep -a "Training progress" | tail -2 | cut -c1-160
python - <<'PY'
import sys; sys.path.insert(0,'tests/e2e_shims')
from plyfile import PlyData
print("gaussians in the saved model:", len(PlyData.read('/tmp/soak_model/point_cloud/iteration_4000/point_cloud.ply')['vertex']))
PY
( time python gaussian-opacity-fields_amd/launch/run_reference_script.py oracle/_ref/refpy/extract_mesh.py -m $M --iteration 4000 ) > gpurun_out/soak/mesh.log 2>&1
grep -a "real\|torch.Size\|binary search in step 7" gpurun_out/soak/mesh.log | cut -c1-160
ls -la $M/test/ours_4000/fusion/ | tail -3
