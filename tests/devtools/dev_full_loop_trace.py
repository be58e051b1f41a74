"""Kernel list of bench.py's full training iteration (rocprofv3 --kernel-trace --stats around this script).
    python tests/devtools/dev_full_loop_trace.py            the three compositions of bench.full_loop
    python tests/devtools/dev_full_loop_trace.py inline     the unchanged train.py composition alone, 20 iterations (round 5: where its ~1.15 ms of glue goes)
    python tests/devtools/dev_full_loop_trace.py launcher   the same lines as the launcher runs them by default (round 6: deferred loss + split SH)
Post-processing of the trace: tests/devtools/dev_trace_summary.py."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("", "gaussian-opacity-fields_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import bench
import synthetic_scenes as S
from gpu_common import to_dev
dev = torch.device("cuda", 0)
sd = to_dev(S.scene_frustum(1_000_000, W=1600, H=1063, focal=1200.0, seed=0), dev)
if len(sys.argv) > 1 and sys.argv[1] == "inline":
    print(bench.full_loop(sd, dev, 1600, 1063, steps=20, warmup=3, only_inline=True))
elif len(sys.argv) > 1 and sys.argv[1] == "launcher":
    print(bench.full_loop(sd, dev, 1600, 1063, steps=20, warmup=3, only_launcher=True))
else:
    print(bench.full_loop(sd, dev, 1600, 1063, steps=10, warmup=2))
