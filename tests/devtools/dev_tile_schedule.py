"""Developer script: would another tile order shorten the blend kernels?  Per-tile work (contributing pairs, visited entries) of the
S1M frame -> greedy list-scheduling makespan of (a) the shipped order (contiguous band of tiles per XCD, ascending), (b) longest-first
inside every band, (c) longest-first with the bands balanced by work, against the perfectly divisible bound."""
import sys, os, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import _backend as B
sc = S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
res = product_forward_raw(sd)
W, H = sd["W"], sd["H"]
gx, gy = (W + 15) // 16, (H + 15) // 16
nt = gx * gy
pairs = fetch(res, "contrib_pairs").astype(np.float64)
ncon = fetch(res, "n_contrib")[: W * H].reshape(H, W)
ranges = fetch(res, "ranges").reshape(-1, 2)
length = (ranges[:, 1] - ranges[:, 0]).astype(np.float64)
visited = np.zeros(nt)
pad = np.zeros((gy * 16, gx * 16), dtype=ncon.dtype); pad[:H, :W] = ncon
visited = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1).astype(np.float64)
print("tiles %d: list length mean %.0f max %.0f; visited mean %.0f max %.0f; pairs mean %.0f max %.0f" % (
    nt, length.mean(), length.max(), visited.mean(), visited.max(), pairs.mean(), pairs.max()))
cost = pairs / 48.0 + 0.5 * visited            # ~ wave iterations of phase 2 + staging / scan share
def makespan(orders, slots):
    # orders: list (per XCD) of tile index arrays in dispatch order; every XCD has `slots` workgroup slots
    worst = 0.0
    for o in orders:
        h = [0.0] * slots
        heapq.heapify(h)
        end = 0.0
        for t in o:
            s = heapq.heappop(h)
            e = s + cost[t]
            end = max(end, e)
            heapq.heappush(h, e)
        worst = max(worst, end)
    return worst
per = (nt + 7) // 8
bands = [np.arange(x * per, min(nt, (x + 1) * per)) for x in range(8)]
for name, slots in (("forward (5 workgroups/CU)", 160), ("backward (6 workgroups/CU)", 192)):
    ideal = cost.sum() / (8 * slots)
    a = makespan(bands, slots)
    b = makespan([o[np.argsort(-cost[o])] for o in bands], slots)
    order = np.argsort(-cost)
    bal = [order[x::8] for x in range(8)]      # deal the tiles out longest-first: balances the bands
    c = makespan(bal, slots)
    print("%s: ideal %.0f | shipped order %.0f (+%.1f %%) | longest-first per band %.0f (+%.1f %%) | longest-first, bands balanced %.0f (+%.1f %%); band sums min %.0f max %.0f" % (
        name, ideal, a, 100 * (a / ideal - 1), b, 100 * (b / ideal - 1), c, 100 * (c / ideal - 1),
        min(cost[o].sum() for o in bands) / slots, max(cost[o].sum() for o in bands) / slots))
