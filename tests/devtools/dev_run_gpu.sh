cd $GRAFT_REPO_ROOT
for g in 200000 5000000; do
python bench.py --gaussians $g --no-cpu-baseline --no-full-loop 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print($g, d['value'], d['ms_per_step'], d['config']['num_rendered'], {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})"
done
python tests/devtools/dev_integrate_cache_bench.py 2>&1 | tail -6
