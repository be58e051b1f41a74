"""Developer script: loop statistics of blend_backward at S1M (library built with -DGOF_STATS)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import _backend as B
sc = S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
res = product_forward_raw(sd)
dL = torch.randn(9, sd["H"], sd["W"], device="cuda")
a = res["args"]
out = (C.c_ulonglong * 12)()
B.lib.gof_debug_bw_stats(out, 1)
B.rasterize_gaussians_backward(a[0], a[1], res["radii"], a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                               dL, a[17], a[18], a[19], res["geom"], res["R"], res["binning"], res["img"], False)
torch.cuda.synchronize()
B.lib.gof_debug_bw_stats(out, 1)
s = list(out)
nwaves = 6700 * 4
print("staged entries %d (per tile %.1f); wave iterations of the entry loop %d (per wave %.1f)" % (s[4], s[4] / 6700, s[0], s[0] / nwaves))
print("contributing (pixel, entry) pairs %d; lane utilisation of the gradient block %.3f" % (s[2], s[2] / max(1, 64.0 * s[0])))
