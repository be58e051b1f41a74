cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_epilogue_gpu.py -m gpu -q -x -k "l1_loss or composition or end_to_end or mirrors" --tb=short 2>&1 | cut -c1-400 | tail -8
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
f=d['full_loop']; print(d['value'], f['ms_per_iter'], f['one_call_loss']['ms_per_iter'], f['one_call_loss_split_sh']['ms_per_iter'])"
