cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_mtets_gpu.py tests/test_knn.py -m gpu -q -x --tb=short 2>&1 | cut -c1-400 | tail -6
python bench.py --no-cpu-baseline --no-full-loop 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); print({k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})"
GOF_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-200
