#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$PWD/tests/e2e_shims
S=/tmp/dp_scene; M=/tmp/dp_model
python tests/fixtures/make_blender_scene.py $S > /dev/null 2>&1
GOF_DEBUG_FILTER=1 GOF_DP_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    gaussian-opacity-fields_amd/launch/run_train_dp.py oracle/_ref/refpy/train.py -s $S -m $M --iterations 5 --test_iterations 1 --save_iterations 5 --eval > gpurun_out/ab7_dp.log 2>&1
grep -a "filter debug" gpurun_out/ab7_dp.log
GOF_DEBUG_FILTER=1 python gaussian-opacity-fields_amd/launch/run_reference_script.py oracle/_ref/refpy/train.py -s $S -m ${M}s --iterations 5 --test_iterations 1 --save_iterations 5 --eval 2>&1 | grep -a "filter debug"
GOF_DEBUG_FILTER=1 python gaussian-opacity-fields_amd/launch/run_reference_script.py oracle/_ref/refpy/train.py -s $S -m ${M}s2 --iterations 5 --test_iterations 1 --save_iterations 5 --eval 2>&1 | grep -a "filter debug"
