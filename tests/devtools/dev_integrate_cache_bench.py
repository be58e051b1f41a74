"""Developer script: integrate with the per-view cache (mesh-extraction pattern): first call of a view vs later calls."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
import diff_gaussian_rasterization as DGR
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B

def run(P, sigma_px, label):
    sc = S.scene_frustum(P, seed=0, sigma_px=sigma_px)
    pts = torch.from_numpy(S.tetra_points(sc)).cuda()
    sd = to_dev(sc)
    r = GaussianRasterizer(settings_from(sd))
    def call(p):
        return r.integrate(points3D=p, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"], scales=sd["scales"], rotations=sd["rotations"])
    call(pts); torch.cuda.synchronize()
    DGR.integrate_view_cache().clear()
    for i, p in enumerate((pts, pts, pts[::3].contiguous(), pts[::20].contiguous())):
        B.profile_enable(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with DGR.integrate_view_key((label, 0)):
            out = call(p)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rep = B.profile_report(); B.profile_enable(False)
        print(f"{label} call {i} ({p.shape[0]} points): wall {dt*1e3:.2f} ms  kernels {({k: round(v['total_ms'], 3) for k, v in rep.items()})}")
    c = DGR.integrate_view_cache()
    print(f"   cache: {c.bytes/2**20:.0f} MiB for 1 view, hits {c.hits} misses {c.misses}")
    c.clear()
    del pts

run(1_000_000, 3.0, "S1M")
run(5_000_000, 1.5, "S5M")
