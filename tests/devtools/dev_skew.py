"""Developer script: a SKEWED S1M scene (70% of the Gaussians projected into the central 25% of the image: per-tile list lengths
differ by ~10x) -- does the static XCD-banded tile order leave a tail?  Prints per-kernel times and the list-length spread."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import _backend as B
for skew in (False, True):
    sc = S.scene_frustum(1_000_000, seed=0)
    if skew:
        rng = np.random.default_rng(5)
        P = sc["means3D"].shape[0]
        m = rng.random(P) < 0.7
        sc["means3D"][m, 0] *= 0.45; sc["means3D"][m, 1] *= 0.45            # pull 70% towards the optical axis
    sd = to_dev(sc)
    res = product_forward_raw(sd)
    ranges = fetch(res, "ranges").view(np.uint32).reshape(-1, 2).astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    dL = torch.randn(9, sd["H"], sd["W"], device="cuda")
    a = res["args"]
    def bwd():
        return B.rasterize_gaussians_backward(a[0], a[1], res["radii"], a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                                              dL, a[17], a[18], a[19], res["geom"], res["R"], res["binning"], res["img"], False)
    for _ in range(3): product_forward_raw(sd); bwd()
    B.profile_enable(True)
    for _ in range(10): product_forward_raw(sd); bwd()
    rep = B.profile_report(); B.profile_enable(False)
    t = {k: round(v["total_ms"] / v["calls"], 4) for k, v in rep.items()}
    print("skew" if skew else "uniform", "R", res["R"], "list len mean %.0f p99 %.0f max %d" % (lens.mean(), np.percentile(lens, 99), lens.max()),
          "fwd %.3f bwd %.3f" % (t["blend_forward"], t["blend_backward"]), "ns per instance fwd %.2f bwd %.2f" % (t["blend_forward"] * 1e6 / res["R"], t["blend_backward"] * 1e6 / res["R"]))
