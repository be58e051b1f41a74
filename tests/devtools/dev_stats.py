"""Developer script: workload statistics of S1M on the GPU (list lengths, walked entries, contributors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
sc = S.scene_frustum(1_000_000, seed=0)
res = product_forward_raw(to_dev(sc)); torch.cuda.synchronize()
W, H = sc["W"], sc["H"]; gx = (W + 15) // 16; gy = (H + 15) // 16
ranges = fetch(res, "ranges").view(np.uint32).reshape(-1, 2).astype(np.int64)
lens = ranges[:, 1] - ranges[:, 0]
nc = fetch(res, "n_contrib").view(np.uint32).reshape(2, H, W).astype(np.int64)
last = nc[0]
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = last
tmax = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).ravel()
wave = pad.reshape(gy, 4, 4, gx, 16).max(axis=(2, 4))   # per wave (4 rows x 16)
print("R", res["R"], "visible", int((res["radii"] > 0).sum()), "tiles", len(lens))
print("list len mean %.1f max %d" % (lens.mean(), lens.max()))
print("last_contributor per pixel mean %.1f" % last.mean())
print("tile max_last mean %.1f  sum %d (%.1f%% of R)" % (tmax.mean(), tmax.sum(), 100.0 * tmax.sum() / res["R"]))
print("wave max_last sum %d  -> wave-entries bwd (tile-level staging) %d, wave-level %d" % (wave.sum(), 4 * tmax.sum(), wave.sum()))
alpha = res["color"][7].cpu().numpy()
print("alpha mean %.3f, saturated(T<1e-3) %.3f" % (alpha.mean(), (alpha > 0.999).mean()))
fT = fetch(res, "final_T").reshape(4, H, W)
print("pixels terminated early (T*(1-a)<1e-4): approx", float((fT[0] < 2e-4).mean()))
