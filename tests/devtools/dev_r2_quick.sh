#!/bin/bash
# quick check after a kernel change: backward / gradient / DP tests, then the headline bench without the side legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_reference_gpu.py tests/test_dp_gpu.py tests/test_train_epilogue_gpu.py -q -x -k "${KEXPR:-backward or gradient or autograd or s1m or reproducible or dp or exchange or training}" 2>&1 | grep -v "ERROR: Maximal" | tail -5
python bench.py --no-cpu-baseline --no-integrate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})
print({k:(v if not isinstance(v,dict) else v.get('ms_per_iter')) for k,v in d.get('full_loop',{}).items() if k in ('ms_per_iter','one_call_loss','one_call_loss_split_sh')})"
