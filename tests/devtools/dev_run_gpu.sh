cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench11.json 2> gpurun_out/bench11.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof11 -o r11 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-loop > $R/gpurun_out/prof11.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc11_$c -o p -- python $R/tests/devtools/dev_pmc.py > $R/gpurun_out/pmc11_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc11_valu -o p -- python $R/tests/devtools/dev_pmc.py > $R/gpurun_out/pmc11_valu.log 2>&1
cd $R; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench11.json'))
print(d["value"], d["ms_per_step"], d["full_loop"]["ms_per_iter"], {k:v["avg_ms"] for k,v in d["roofline"]["kernels"].items()})
PY
