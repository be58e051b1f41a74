"""Developer script: per-kernel times of the Gaussian-side stages (preprocess, depth sort, scan, emission, tile sort) and the backward's
per-Gaussian stages at S1M and at 6M Gaussians @ 1237x822 (library selected with GOF_HIP_LIB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
for label, sc in (("S1M", S.scene_frustum(1_000_000, seed=0)), ("6M@1237x822", S.scene_frustum(6_000_000, W=1237, H=822, focal=1237.0 * 0.75, seed=0, sigma_px=1.5))):
    sd = to_dev(sc)
    params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    rast = GaussianRasterizer(settings_from(sd))
    dL = torch.randn((9, sd["H"], sd["W"]), device="cuda")
    def step():
        for p in params.values(): p.grad = None
        color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
        color.backward(dL)
    for _ in range(4): step()
    torch.cuda.synchronize()
    B.profile_enable(True)
    for _ in range(6): step()
    torch.cuda.synchronize()
    rep = B.profile_report(); B.profile_enable(False)
    print(label, {k: round(v["total_ms"] / v["calls"], 4) for k, v in rep.items() if not k.startswith("blend")}, flush=True)
    del sd, params, means2D, rast, dL
    torch.cuda.empty_cache()
