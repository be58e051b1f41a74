#!/bin/bash
# Round 6, GPU call D: the three-pass depth sort under the binding's key promise (GofRasterArgs.depth_key_bits = 27) against the four-pass
# sort of the same library (SHORT_DEPTH_KEYS=False), interleaved; first the parity subset that exercises it.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06d
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x --tb=short -k "forward_bit_exact or fused_forward or full_size_s1m_against_oracle or large_p or two_streams or channel_slices or learnt_mask_pool or backward_blend_gradients" > $O/tests.txt 2>&1; tail -8 $O/tests.txt | cut -c1-800
timeout 600 python tests/devtools/dev_r6_ab.py three_pass: four_pass::SHORT_DEPTH_KEYS=False > $O/ab.txt 2> $O/ab.err; tail -2 $O/ab.err; cut -c1-600 $O/ab.txt
timeout 300 python bench.py --no-cpu-baseline --no-integrate --no-clustered --no-views --no-reference --no-kernel-size-leg --no-large-p > $O/bench_short.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r06d/bench_short.json"))
print(d["value"], d["ms_per_step"], {k: d["full_loop"][k] for k in d["full_loop"] if k in ("iters_per_s", "ms_per_iter")}, d["full_loop"].get("launcher_default"), d["full_loop"].get("one_call_loss_split_sh"))
PY
