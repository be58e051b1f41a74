"""Developer script (round 6, GPU): does the number of streams a process created BEFORE the library's first forward change the step time?
(The library's second stream is the (n + 1)-th stream of the process; HIP deals hardware queues round-robin.)  One subprocess per n.
    python tests/devtools/dev_r6_streams_before.py            -> a line per n = 0..6
    python tests/devtools/dev_r6_streams_before.py <n>        -> the timing of one process"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for n in range(7):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n)], capture_output=True, text=True)
        print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
    sys.exit(0)
import torch  # noqa: E402
from gpu_common import to_dev, settings_from  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B  # noqa: E402

n = int(sys.argv[1])
keep = []
for _ in range(n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device="cuda") + 1)
torch.cuda.synchronize()
sd = to_dev(S.scene_frustum(1_000_000, seed=0))
params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
means2D = torch.zeros_like(params["means3D"], requires_grad=True)
rast = GaussianRasterizer(settings_from(sd))
dL = torch.randn((9, sd["H"], sd["W"]), device="cuda")


def step():
    for p in params.values():
        p.grad = None
    color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dL)


for _ in range(5):
    step()
best = 1e9
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
B.profile_enable(True)
for _ in range(5):
    step()
rep = B.profile_report(); B.profile_enable(False)
k = {a: round(v["total_ms"] / max(1, v["calls"]), 4) for a, v in rep.items() if a in ("preprocess_fwd", "preprocess_fwd_heavy", "sort_gaussians_by_depth")}
print("streams created before the first forward: %d   S1M step %.4f ms   %s" % (n, best, k))
