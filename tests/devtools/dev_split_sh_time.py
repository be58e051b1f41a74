"""Per-kernel times of forward + backward at S1M with the SH coefficients concatenated vs as two tensors (SplitSH)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("gaussian-opacity-fields_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import synthetic_scenes as S
from gpu_common import to_dev, settings_from
from diff_gaussian_rasterization import GaussianRasterizer, SplitSH, _backend as B
dev = torch.device("cuda", 0)
sd = to_dev(S.scene_frustum(1_000_000, W=1600, H=1063, focal=1200.0, seed=0), dev)
dL = torch.randn((9, 1063, 1600), generator=torch.Generator().manual_seed(1)).to(dev)
rast = GaussianRasterizer(settings_from(sd))
for mode in ("cat", "split", "cat", "split"):
    leaves = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations")}
    dc = sd["shs"][:, :1].clone().requires_grad_(True); rest = sd["shs"][:, 1:].clone().requires_grad_(True)
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    def step():
        for t in list(leaves.values()) + [dc, rest, m2]:
            t.grad = None
        shs = torch.cat((dc, rest), 1) if mode == "cat" else SplitSH(dc, rest)
        c, _ = rast(means3D=leaves["means3D"], means2D=m2, shs=shs, opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"])
        c.backward(dL)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    B.profile_enable(True)
    for _ in range(5):
        step()
    r = B.profile_report(); B.profile_enable(False)
    print(mode, "wall %.3f ms" % wall, {k: round(v["total_ms"] / v["calls"], 4) for k, v in r.items() if k in ("preprocess_fwd", "preprocess_bwd", "blend_forward", "blend_backward")})
