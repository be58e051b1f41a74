#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tests/devtools/microbench/valu_rate 2>&1 | tee gpurun_out/ab4_valu_rate.txt
export PYTHONPATH=$PWD/tests/e2e_shims
S=/tmp/dp_scene; M=/tmp/dp_model
python tests/fixtures/make_blender_scene.py $S > /dev/null 2>&1
GOF_DP_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  gaussian-opacity-fields_amd/launch/run_train_dp.py oracle/_ref/refpy/train.py -s $S -m $M --iterations 400 --densify_from_iter 100 --densification_interval 100 \
  --opacity_reset_interval 300 --densify_until_iter 900 --test_iterations 1 100 400 --save_iterations 400 --eval > gpurun_out/ab4_dp.log 2>&1
echo "dp rc=$?"
grep -v "Training progress" gpurun_out/ab4_dp.log | grep -i -B2 -A12 "rank1\]\|terminate\|what()\|Abort\|core dumped" | head -80
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py -x -q 2>&1 | tail -25 | tee gpurun_out/ab4_pytest.txt
