"""Developer script: integrate at S1M (1M Gaussians, 9M query points) -- product vs the REFERENCE's own integrateCUDA
(oracle/_ref, hipcc build), meant to be run under rocprofv3 --kernel-trace --stats; also a phase-A-only product call
(9 query points) to split the kernel time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S, reference_binding as rb
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sig = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
sc = S.scene_frustum(P, seed=0, sigma_px=sig)
pts_np = S.tetra_points(sc)
sd = to_dev(sc)
pts = torch.from_numpy(pts_np).cuda()
r = GaussianRasterizer(settings_from(sd))
def call(p):
    return r.integrate(points3D=p, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"], scales=sd["scales"], rotations=sd["rotations"])
for p, tag in ((pts, "full"), (pts[:9].contiguous(), "phaseA-only (9 points)")):
    call(p); torch.cuda.synchronize()
    B.profile_enable(True)
    t0 = time.perf_counter(); out = call(p); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rep = B.profile_report(); B.profile_enable(False)
    print("product %s: wall %.2f ms; kernels %s" % (tag, dt * 1e3, {k: round(v["total_ms"], 3) for k, v in rep.items()}))
if rb.available(""):
    ref = rb.Reference(sd, "")
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); res = ref.integrate(pts_np); t1 = time.perf_counter()
        print("reference integrate wall (incl. H2D/D2H of %d points): %.1f ms" % (pts_np.shape[0], (t1 - t0) * 1e3))
    a_ref = res[1] if isinstance(res, tuple) else None
    if a_ref is not None:
        a = out[1].cpu().numpy() if tag == "full" else call(pts)[1].cpu().numpy()
        d = np.abs(a - a_ref)
        print("alpha_integrated product vs reference: max %.3g, frac > 1e-3: %.3g" % (d.max(), (d > 1e-3).mean()))
