cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_epilogue_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "training_loss or backward or gradient or determin or bucket or end_to_end" --tb=short 2>&1 | cut -c1-300 | tail -5
python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['kernels']['backward_memsets'])
print(d['full_loop']['ms_per_iter'], d['full_loop']['one_call_loss']['ms_per_iter'])"
