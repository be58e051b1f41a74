"""Developer script: integrate (opacity-field query) at scale -- timing + size-independent properties."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B

def run(P, sigma_px, label):
    sc = S.scene_frustum(P, seed=0, sigma_px=sigma_px)
    pts = torch.from_numpy(S.tetra_points(sc)).cuda()
    sd = to_dev(sc)
    r = GaussianRasterizer(settings_from(sd))
    def call():
        return r.integrate(points3D=pts, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"],
                           scales=sd["scales"], rotations=sd["rotations"])
    call(); torch.cuda.synchronize()
    B.profile_enable(True)
    t0 = time.perf_counter(); color, alpha, colp, radii = call(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rep = B.profile_report(); B.profile_enable(False)
    inside = alpha != 1.0
    n_in = int(color[8].sum().item())
    print(f"{label}: P={P} PN={pts.shape[0]} wall {dt*1e3:.1f} ms  Mpoints/s {pts.shape[0]/dt/1e6:.1f}  points in image {n_in}")
    print("   kernels:", {k: round(v['total_ms'], 3) for k, v in rep.items()})
    a = alpha.cpu().numpy()
    assert np.isfinite(a).all() and a.min() >= 0 and a.max() <= 1.0 + 1e-5
    assert (a < 1.0).sum() <= n_in                          # only points inside the image are written
    c2, a2, _, _ = call()
    assert torch.equal(a2, alpha) and torch.equal(c2, color)    # idempotent / deterministic
    del pts

run(1_000_000, 3.0, "S1M")
run(5_000_000, 1.5, "S5M (BASELINE config 5)")
