"""Developer script (round 6, GPU): fwd+bwd steps of bench.py's `large_p` scene bicycle_like_6M (6M Gaussians @ 1237x822, sigma_px 1.5) through
the autograd surface, for a rocprofv3 kernel trace of the regime of real captures:
    rocprofv3 --kernel-trace --stats -d <dir> -- python tests/devtools/dev_r6_large_p_trace.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpu_common import to_dev, settings_from  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sd = to_dev(S.scene_frustum(6_000_000, W=1237, H=822, focal=1237.0 * 0.75, seed=0, sigma_px=1.5))
params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
means2D = torch.zeros_like(params["means3D"], requires_grad=True)
rast = GaussianRasterizer(settings_from(sd))
dL = torch.randn((9, sd["H"], sd["W"]), device="cuda")
for i in range(steps + 4):
    if i == 4:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    for p in params.values():
        p.grad = None
    color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dL)
e1.record()
torch.cuda.synchronize()
print("bicycle_like_6M: %.4f ms per fwd+bwd step over %d steps" % (e0.elapsed_time(e1) / steps, steps))
