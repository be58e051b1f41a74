cd $GRAFT_REPO_ROOT
GOF_BENCH_SHARE_GPU=nccl NCCL_DEBUG=WARN timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 2 --steps 4 --warmup 2 2>&1 | grep -v "^\s*$" | grep -i -B2 -A12 "error\|Traceback" | head -60 | cut -c1-250
