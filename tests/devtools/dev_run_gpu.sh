cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py -m gpu -q -x -k "not full_size_s1m_against and not config5" 2>&1 | tail -1
for P in 1000000 5000000; do timeout 600 python bench.py --steps 10 --warmup 3 --gaussians $P --no-cpu-baseline --no-full-loop 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items() if k in ('preprocess_fwd','scan_tiles','emit_instances')})"; done
