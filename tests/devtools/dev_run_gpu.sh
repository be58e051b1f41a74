cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
GOF_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 4 --warmup 1 --gaussians 200000 2>&1 | grep -E '^\{"metric"' | cut -c1-200
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench8.json 2> gpurun_out/bench8.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof8 -o r8 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-full-loop > $R/gpurun_out/prof8.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc8_$c -o p -- python $R/tests/devtools/dev_pmc.py > $R/gpurun_out/pmc8_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc8_valu -o p -- python $R/tests/devtools/dev_pmc.py > $R/gpurun_out/pmc8_valu.log 2>&1
cd $R; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench8.json'))
print(d["value"], d["ms_per_step"], d["full_loop"]["ms_per_iter"], {k:v["avg_ms"] for k,v in d["roofline"]["kernels"].items()})
PY
