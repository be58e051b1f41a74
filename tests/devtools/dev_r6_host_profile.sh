#!/bin/bash
# Round 6 (GPU): where does the HOST time of a training iteration of the unchanged train.py go on a config-3-shaped scene (the trajectory's: 64 views of 800x800,
# 100k -> 540k Gaussians)?  The GPU work of such an iteration is ~1.5 ms, the loop runs at 4 ms per iteration: cProfile of 3000 iterations through the launcher.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06host; rm -rf $O; mkdir -p $O
W=/tmp/gof_hostprof; rm -rf $W; mkdir -p $W
export PYTHONPATH=$GRAFT_REPO_ROOT/tests/e2e_shims:$PYTHONPATH GOF_E2E_SEED=0 PYTHONUNBUFFERED=1
python tests/fixtures/make_blender_scene.py $W/scene --views 64 --test-views 8 --size 800 800 --gt 400000 --gt-scale 0.007 --init 100000 --large > $O/scene.txt 2>&1
ARGS="-s $W/scene -m $W/model --iterations 3000 --densify_until_iter 2600 --distortion_from_iter 1500 --depth_normal_from_iter 1500 --densify_grad_threshold 0.00012 --test_iterations 3000 --save_iterations 3000 --eval --quiet"
( time python gaussian-opacity-fields_amd/launch/run_reference_script.py oracle/_ref/refpy/train.py $ARGS ) > $O/plain.txt 2>&1; tail -4 $O/plain.txt
rm -rf $W/model
python -m cProfile -o $O/train.prof gaussian-opacity-fields_amd/launch/run_reference_script.py oracle/_ref/refpy/train.py $ARGS > $O/prof_run.txt 2>&1; tail -2 $O/prof_run.txt
python - <<'PY' > $O/top.txt 2>&1
import pstats
p = pstats.Stats("gpurun_out/r06host/train.prof")
p.sort_stats("cumulative").print_stats(70)
p.sort_stats("tottime").print_stats(45)
PY
cut -c1-190 $O/top.txt | head -190
