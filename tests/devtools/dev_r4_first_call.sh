#!/bin/bash
# First GPU call of round 4 (round 3 ended with no GPU minutes left: the cull margins of csrc/preprocess.hip / blend_forward.hip were
# tightened to their derived bounds and validated from source on the host only).  One call, ~12 GPU-minutes:
#   1. the forward-facing part of the GPU suite incl. the cull audit on the shipped constants;
#   2. A/B of the S1M + clustered bench legs: shipped constants | round 2's margins | -DGOF_TIGHT_RECTS (variant libraries are built
#      here if they did not travel; selected with GOF_HIP_LIB);
#   3. the default bench line.
# Then `dev_r3_profiles.sh` (fresh rocprofv3 / PMC pass: blend_forward's code changed by a constant) in a second call.
#   gpurun --timeout 1500 -- 'bash tests/devtools/dev_r4_first_call.sh'
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_first; mkdir -p $O
PKG=gaussian-opacity-fields_amd
[ -f $PKG/lib/libgof_hip_r2margins.so ] || GOF_BUILD_TAG=r2margins GOF_EXTRA_FLAGS="-DGOF_BOX_C=6e-6 -DGOF_CONE_MARGIN=3e-6f" python $PKG/build.py > $O/build_r2margins.log 2>&1
[ -f $PKG/lib/libgof_hip_tight.so ] || GOF_BUILD_TAG=tight GOF_EXTRA_FLAGS="-DGOF_TIGHT_RECTS" python $PKG/build.py > $O/build_tight.log 2>&1
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "forward or audit or integrate_bit_exact or fuzz or full_size" > $O/pytest_forward.txt 2>&1; tail -3 $O/pytest_forward.txt
for v in shipped r2margins tight; do
  lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip.so; [ $v = shipped ] || lib=$GRAFT_REPO_ROOT/$PKG/lib/libgof_hip_$v.so
  for rep in 1 2; do
    GOF_HIP_LIB=$lib timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-integrate --no-full-loop > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err || tail -2 $O/bench_${v}_$rep.err
  done
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r4_first")
for f in sorted(glob.glob(O + "/bench_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["roofline"]["kernels"]
        print(os.path.basename(f), "it/s %.1f ms %.3f | blend_forward %.3f blend_backward %.3f binning %.3f | clustered ms %s" % (
            d["value"], d["ms_per_step"], k["blend_forward"]["avg_ms"], k["blend_backward"]["avg_ms"],
            sum(v["avg_ms"] * v["calls"] for n, v in k.items() if n in ("emit_instances", "sort_tiles", "tile_ranges", "scan_tiles", "sort_gaussians_by_depth", "order_tiles")) / max(k["blend_forward"]["calls"], 1),
            d.get("clustered", {}).get("ms_per_step")))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-300
