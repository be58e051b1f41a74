"""Kernel times of the compressed SH-gradient exchange at the bench size (1M Gaussians), n_views = 2, 4, 8."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("gaussian-opacity-fields_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
from diff_gaussian_rasterization import _backend as B
P, M = 1_000_000, 16
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
means = torch.randn(P, 3, generator=g).to(dev)
geom = torch.zeros(B.lib.gof_geom_bytes(P), dtype=torch.uint8, device=dev)
src = {"P": P, "M": M, "degree": 3, "means3D": means, "campos": torch.zeros(3, device=dev), "dL_dcolors": torch.randn(P, 3, generator=g).to(dev),
       "geom": geom, "radii": torch.ones(P, dtype=torch.int32, device=dev)}
out = torch.empty(P, M, 3, device=dev)
row = torch.empty(P + 1, 3, device=dev)
for n in (2, 4, 8):
    gathered = torch.randn(n, P + 1, 3, generator=g).to(dev)
    for _ in range(3):
        B.sh_grad_pack(src, row); B.sh_grad_expand(src, gathered, 1.0, [out])
    torch.cuda.synchronize()
    B.profile_enable(True)
    for _ in range(10):
        B.sh_grad_pack(src, row); B.sh_grad_expand(src, gathered, 1.0, [out])
    r = B.profile_report(); B.profile_enable(False)
    print(n, {k: round(v["total_ms"] / v["calls"], 4) for k, v in r.items()})
