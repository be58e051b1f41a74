cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "forward_bit or properties_s1m or fused" 2>&1 | tail -1
for P in 1000000 5000000; do timeout 600 python bench.py --steps 10 --warmup 3 --gaussians $P --no-cpu-baseline --no-full-loop 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items() if k in ('scan_tiles','emit_instances','sort_gaussians_by_depth')})"; done
