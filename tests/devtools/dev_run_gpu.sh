cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
print(json.dumps(d['full_loop'],indent=1))"
