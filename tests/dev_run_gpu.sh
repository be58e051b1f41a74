cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GOF_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --gaussians 300000 2>&1 | grep -v "amdgpu.ids" | tail -5 | cut -c1-600
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q -x -k "backward or gradient or determin" 2>&1 | tail -2
