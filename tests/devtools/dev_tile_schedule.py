"""Developer script: what the tile scheduler is worth.  For S1M (uniform tile lists) and S1M-clustered (heavy-tailed: lists of
1300 ... 14500 entries, 220 ... 2060 walked) measure, with the library selected by GOF_HIP_LIB:
  * per-kernel times of forward + backward (HIP events on the launch stream, gof_profile_*);
  * with an instrumented build (-DGOF_STATS: per-tile start / end on the 100 MHz counter): the blend kernels' makespan against
    the work-proportional ideal = sum of the tile durations / number of concurrently resident workgroups (measured as the time
    integral of the number of running tiles / makespan) -- "efficiency" = ideal / makespan.
Prints one JSON line per scene.  Run once per library variant:
    python tests/devtools/dev_tile_schedule.py                                  (product: dynamic longest-first queues)
    GOF_HIP_LIB=.../libgof_hip_static.so python tests/devtools/dev_tile_schedule.py   (-DGOF_STATIC_TILES: round 2's static map)
    GOF_HIP_LIB=.../libgof_hip_audit.so  python tests/devtools/dev_tile_schedule.py   (per-tile clocks)"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gpu_common import fetch, product_forward_raw, to_dev  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from diff_gaussian_rasterization import _backend as B  # noqa: E402

SCENES = {"s1m": lambda: S.scene_frustum(1_000_000, seed=0), "s1m_clustered": lambda: S.scene_clustered(1_000_000, seed=0)}


def tile_stats(fn, ntiles, order=None, qlen=None):
    out = (C.c_ulonglong * (2 * ntiles))()
    assert fn(out, ntiles) == 0
    a = np.frombuffer(out, dtype=np.uint64).astype(np.int64)
    st, en = a[:ntiles], a[ntiles:]
    ok = en > 0
    st, en = st[ok], en[ok]
    dur = (en - st).astype(np.float64)
    makespan = float(en.max() - st.min())
    busy = float(dur.sum())
    # average number of concurrently running tiles = busy / makespan; ideal makespan with perfect packing into the peak concurrency
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    peak = int(np.cumsum(ev[:, 1]).max())
    extra = {}
    grid = np.linspace(st.min(), en.max(), 21)[1:-1]
    extra["running_tiles_at_5pct_steps"] = [int(((st <= g) & (en > g)).sum()) for g in grid]
    if order is not None and ok.all():
        stride = (ntiles + 7) // 8 + 128
        xcd_of = np.zeros(ntiles, dtype=np.int64)
        for x in range(8):
            xcd_of[order[x * stride:x * stride + int(qlen[x])]] = x
        extra["queue_finish_us"] = [round(float(en[xcd_of == x].max() - st.min()) / 100, 1) for x in range(8)]
        extra["queue_busy_us"] = [round(float(dur[xcd_of == x].sum()) / 100, 1) for x in range(8)]
    top = np.argsort(-dur)[:5]
    t0 = st.min()
    ids = np.nonzero(ok)[0]
    return {**extra, "top5_tile_start_us_dur_us": [[int(ids[i]), round(float(st[i] - t0) / 100, 1), round(float(dur[i]) / 100, 1)] for i in top],
            "tiles": int(ok.sum()), "makespan_us": makespan / 100.0, "sum_tile_us": busy / 100.0, "peak_concurrent_tiles": peak,
            "ideal_us": busy / peak / 100.0, "efficiency": busy / peak / makespan, "longest_tile_us": float(dur.max()) / 100.0,
            "mean_tile_us": float(dur.mean()) / 100.0}


def main():
    names = [a for a in sys.argv[1:] if a in SCENES] or list(SCENES)
    for name in names:
        sd = to_dev(SCENES[name]())
        res = product_forward_raw(sd)
        dL = torch.randn(9, sd["H"], sd["W"], device="cuda")
        a = res["args"]

        def bwd():
            return B.rasterize_gaussians_backward(a[0], a[1], res["radii"], a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                                                  dL, a[17], a[18], a[19], res["geom"], res["R"], res["binning"], res["img"], False)
        for _ in range(3):
            product_forward_raw(sd); bwd()
        B.profile_enable(True)
        for _ in range(10):
            product_forward_raw(sd); bwd()
        rep = B.profile_report(); B.profile_enable(False)
        out = {"scene": name, "lib": os.path.basename(os.environ.get("GOF_HIP_LIB", "libgof_hip.so")), "R": int(res["R"]),
               "kernels_ms": {k: round(v["total_ms"] / v["calls"], 4) for k, v in rep.items()}}
        out["fwd_bwd_ms"] = round(sum(out["kernels_ms"].values()), 4)
        rg = fetch(res, "ranges").reshape(-1, 2)
        L = (rg[:, 1].astype(np.int64) - rg[:, 0])
        out["list_length_pct_0_50_90_99_100"] = [int(x) for x in np.percentile(L, [0, 50, 90, 99, 100])]
        cost = fetch(res, "tile_cost").astype(np.int64)
        out["walked_pct_0_50_90_99_100"] = [int(x) for x in np.percentile(cost, [0, 50, 90, 99, 100])]
        if hasattr(B.lib, "gof_debug_fw_tile_clock"):
            nt = len(L)
            product_forward_raw(sd); torch.cuda.synchronize()
            out["blend_forward_tiles"] = tile_stats(B.lib.gof_debug_fw_tile_clock, nt, fetch(res, "tile_order").astype(np.int64), fetch(res, "tile_queue")[8:16])
            bwd(); torch.cuda.synchronize()
            out["blend_backward_tiles"] = tile_stats(B.lib.gof_debug_bw_tile_clock, nt, fetch(res, "tile_order_bw").astype(np.int64), fetch(res, "tile_queue")[40:48])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
