cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "two_ranks or sh_gradient or split_sh or share_one or determin" --tb=short 2>&1 | cut -c1-400 | tail -6
GOF_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 6 --warmup 2 2>&1 | tail -1 | cut -c1-330
python bench.py --no-cpu-baseline --no-full-loop 2>&1 | tail -1 | cut -c1-200
