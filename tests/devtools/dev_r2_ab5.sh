#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$PWD/tests/e2e_shims
S=/tmp/dp_scene; M=/tmp/dp_model
python tests/fixtures/make_blender_scene.py $S > /dev/null 2>&1
for mode in default dense; do
  extra=""; [ $mode = dense ] && extra="GOF_DP_DENSE_SH=1"
  env $extra GOF_DP_CHECK_EVERY=1 GOF_DP_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    gaussian-opacity-fields_amd/launch/run_train_dp.py oracle/_ref/refpy/train.py -s $S -m $M$mode --iterations 1150 --densify_from_iter 100 --densification_interval 100 \
    --opacity_reset_interval 300 --densify_until_iter 900 --distortion_from_iter 600 --depth_normal_from_iter 600 --test_iterations 1 100 1150 --save_iterations 1150 --eval > gpurun_out/ab5_dp_$mode.log 2>&1
  echo "dp $mode rc=$?"
  grep -v "Training progress" gpurun_out/ab5_dp_$mode.log | grep -i "diverged\|terminate\|what()\|Abort\|Error" | head -8
done
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py -q 2>&1 | tail -40 | tee gpurun_out/ab5_pytest.txt
