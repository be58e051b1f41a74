#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$PWD/gaussian-opacity-fields_amd/lib
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py -q -x 2>&1 | grep -v "ERROR: Maximal" | tail -30 | tee gpurun_out/ab6_pytest.txt
for v in "" _b128 _r1; do
  echo "== variant '$v'"
  GOF_HIP_LIB=$L/libgof_hip$v.so timeout 300 python tests/devtools/dev_time.py 2>&1 | tail -2
done 2>&1 | tee gpurun_out/ab6_time.txt
GOF_HIP_LIB=$L/libgof_hip_stats.so timeout 300 python tests/devtools/dev_bwstats.py 2>&1 | tail -2 | tee gpurun_out/ab6_bwstats.txt
export PYTHONPATH=$PWD/tests/e2e_shims
S=/tmp/dp_scene; M=/tmp/dp_model
python tests/fixtures/make_blender_scene.py $S > /dev/null 2>&1
GOF_DP_CHECK_EVERY=1 GOF_DP_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    gaussian-opacity-fields_amd/launch/run_train_dp.py oracle/_ref/refpy/train.py -s $S -m $M --iterations 30 --test_iterations 1 --save_iterations 30 --eval > gpurun_out/ab6_dp.log 2>&1
echo "dp rc=$?"
grep -a -v "Training progress" gpurun_out/ab6_dp.log | grep -a -i "diverged\|Evaluating" | head -8
