"""Developer script: host time of one fwd+bwd step through the autograd surface -- a scene so small that the GPU is idle most of the
time (1000 Gaussians @ 64x48): ms per step ~ Python + ctypes + launch overhead + the forward's one event wait.  cProfile of 300 steps."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
sc = S.scene_frustum(1000, W=64, H=48, focal=50.0, seed=3)
sd = to_dev(sc)
params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
means2D = torch.zeros_like(params["means3D"], requires_grad=True)
rast = GaussianRasterizer(settings_from(sd))
dL = torch.randn((9, sd["H"], sd["W"]), device="cuda")
def step():
    for p in params.values(): p.grad = None
    means2D.grad = None
    color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dL)
for _ in range(20): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300): step()
    torch.cuda.synchronize()
    print("host-bound step: %.1f us" % (1e6 * (time.perf_counter() - t0) / 300), flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
