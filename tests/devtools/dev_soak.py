"""Developer script: soak -- many forward/backward iterations at S1M and many small sorts; every forward must reproduce the first
image bit for bit (the single-kernel radix passes poll other workgroups: any ordering bug shows up as a changed list)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer
from simple_knn._C import distCUDA2
sc = S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
means2D = torch.zeros_like(params["means3D"], requires_grad=True)
rast = GaussianRasterizer(settings_from(sd))
dL = torch.randn((9, sd["H"], sd["W"]), device="cuda")
ref = None
t0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for it in range(N):
    for p in params.values(): p.grad = None
    color, radii = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dL)
    if ref is None: ref = color.detach().clone()
    elif it % 20 == 0: assert torch.equal(color.detach(), ref), "forward changed at iteration %d" % it
torch.cuda.synchronize()
print("%d iterations, %.2f ms each, image identical throughout" % (N, (time.perf_counter() - t0) / N * 1e3))
pts = torch.rand((1_500_000, 3), device="cuda")
first = distCUDA2(pts)
for it in range(60):
    assert torch.equal(distCUDA2(pts), first)
print("60 knn runs identical")
