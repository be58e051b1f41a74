"""Developer script: per-call timing of forward / backward at S1M with HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import _backend as B
# usage: dev_time.py [s1m|clustered|posed]   (GOF_HIP_LIB selects a variant library, GOF_FW_EXACT=1 the forward's verification mode)
which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
sc = {"s1m": lambda: S.scene_frustum(1_000_000, seed=0), "clustered": lambda: S.scene_clustered(1_000_000, seed=0),
      "posed": lambda: S.scene_frustum(1_000_000, seed=0, pose_seed=21)}[which]()
sd = to_dev(sc)
res = product_forward_raw(sd)
dL = torch.randn(9, sd["H"], sd["W"], device="cuda")
a = res["args"]
def bwd():
    return B.rasterize_gaussians_backward(a[0], a[1], res["radii"], a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                                          dL, a[17], a[18], a[19], res["geom"], res["R"], res["binning"], res["img"], False)
def timeit(fn, n=10):
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return np.median(ts)
for _ in range(3): bwd(); product_forward_raw(sd)
print(which, "instances", int(res["R"]))
print("fwd ms", timeit(lambda: product_forward_raw(sd)), "bwd ms", timeit(bwd))
B.profile_enable(True)
for _ in range(10): product_forward_raw(sd); bwd()
rep = B.profile_report(); B.profile_enable(False)
print({k: round(v["total_ms"] / v["calls"], 4) for k, v in rep.items()})
