#!/bin/bash
# full validation pass: GPU test suite, smoke, the default bench line, available PMC counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "ERROR: Maximal" | tail -25 | tee gpurun_out/full_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
( time python bench.py ) > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; tail -3 gpurun_out/full_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['roofline']['kernels'].items()})
print(d['roofline']['valu']); print(d.get('integrate')); print(d.get('cpu_baseline')); print({k:(v if not isinstance(v,dict) else v.get('ms_per_iter')) for k,v in d.get('full_loop',{}).items()})
PY
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>&1 | grep -i "SQ_ACTIVE_INST\|SQ_THREAD_CYCLES\|SQ_INSTS_VALU\|SQ_WAVE_CYCLES\|SQ_BUSY_CY\|SQ_WAIT_INST\|SQ_INST_CYCLES\|LDS_BANK\|SQ_LDS\|VALUBusy\|VALUUtil" | cut -c1-200 | sort -u | head -60) > gpurun_out/full_counters.txt 2>&1
wc -l gpurun_out/full_counters.txt
