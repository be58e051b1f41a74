#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json: train iters/s (forward + backward of the
rasterizer, one view per step) on the synthetic S1M scene (1M Gaussians, 1600x1063, SH degree 3),
N GPUs data-parallel over views with an RCCL exchange of the Gaussian parameter gradients per step (the SH gradient as an
all-gather of 12 B per Gaussian and view expanded locally, the other 44 B all-reduced; dp/reducer.py).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- achieved vs peak HBM bandwidth of the dominant kernel, timed with HIP events
  cpu_baseline -- the oracle (CPU restatement of the reference) on a bounded sample, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402


PMC_FILE = "r06_pmc_traffic_s1m.json"
FLOP_PER_PAIR_FWD = 85       # forward.cu:504-575 per contributing pair
FLOP_PER_PAIR_BWD = 190      # backward.cu:771-952 per contributing pair, incl. its 19 accumulating adds

# which translation unit a profiled kernel (its GOF_PROFILE scope / its name in a rocprofv3 trace) is compiled from: a committed
# counter pass is quoted for a kernel only while the hash of THAT file + the shared headers is the one the pass was collected on
KERNEL_SOURCES = {
    "blend_forward": "blend_forward.hip", "blend_forward_exact": "blend_forward.hip",
    "blend_backward": "blend_backward.hip", "gather_tile_partials": "blend_backward.hip",
    "preprocess_fwd": "preprocess.hip", "preprocess_fwd_heavy": "preprocess.hip", "preprocess_bwd": "preprocess.hip", "preprocess_points": "preprocess.hip",
    "emit_instances": "binning.hip", "tile_ranges": "binning.hip", "order_tiles": "binning.hip", "gather_scan_rects": "binning.hip",
    "point_keys": "binning.hip", "gather_sorted_points": "binning.hip",
    "os_hist": "radix.hip", "os_pass": "radix.hip", "rs_hist": "radix.hip", "rs_scatter": "radix.hip", "scan_block": "radix.hip",
    "integrate_pixels": "integrate.hip", "integrate_points": "integrate.hip", "integrate_rays": "integrate.hip", "integrate_pixels_capped": "integrate.hip",
    "rot3_apply_kernel": "gaussian_model_ops.hip",
}
SHARED_HEADERS = ("gof_common.h", "gof_status.h")


def kernel_sha16(kernel=None):
    """Hash of the source a kernel is compiled from (its .hip file + the shared headers) -- or, without `kernel`, of the two blend
    kernels' files: identifies what a committed PMC pass was collected on."""
    import hashlib
    h = hashlib.sha256()
    files = ("blend_forward.hip", "blend_backward.hip") if kernel is None else (KERNEL_SOURCES.get(kernel, kernel + ".hip"),)
    for f in tuple(files) + SHARED_HEADERS:
        path = os.path.join(ROOT, "gaussian-opacity-fields_amd", "csrc", f)
        if os.path.exists(path):
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_pass_is_current(pmc_all, kernel):
    """Was the committed counter pass collected on the code `kernel` has now?  The recorded source hash of that kernel's translation
    unit (+ shared headers) is the current one, or the current one is listed as a later source state with the SAME default-build
    gfx950 code (edits under developer-only #ifdefs; established by tests/devtools/dev_same_isa.py)."""
    cur = kernel_sha16(kernel)
    return cur in [pmc_all.get("_sha16_by_kernel", {}).get(kernel)] + list(pmc_all.get("_same_isa_sha16_by_kernel", {}).get(kernel, []))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1063)
    ap.add_argument("--kernel-size", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-loop", action="store_true", help="skip the full training-iteration leg (losses + Adam)")
    ap.add_argument("--cpu-baseline-gaussians", type=int, default=1_000_000, help="Gaussians of the cpu_baseline sample (default: the whole workload)")
    ap.add_argument("--no-integrate", action="store_true", help="skip the opacity-field query leg (BASELINE config 5 shape)")
    ap.add_argument("--no-clustered", action="store_true", help="skip the heavy-tailed scene leg (S1M-clustered: what the tile scheduler is for)")
    ap.add_argument("--no-views", action="store_true", help="skip the leg that cycles 8 posed cameras inside its timed region")
    ap.add_argument("--no-reference", action="store_true", help="skip the leg that times the reference's own kernels (oracle/_ref) on this GPU")
    ap.add_argument("--no-kernel-size-leg", action="store_true", help="skip the leg that times the headline scene with kernel_size 0.1")
    ap.add_argument("--no-large-p", action="store_true", help="skip the multi-million-Gaussian legs (6M @ 1237x822, 5M @ 1600x1063: fwd+bwd and the full iteration)")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (RANK unset): re-run this command line under torch.distributed.run with one
    process per GPU (the same line the driver uses for N > 1) and hand its exit code back.  The ranks find RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment and take the branch below."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL over dmabuf IPC (the host driver supports nothing else)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); start it as `python bench.py --gpus N` "
                         "or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    distributed = world > 1
    # GOF_BENCH_SHARE_GPU=1 (development only): run the N-rank flow on fewer GPUs than ranks, with gloo instead of RCCL
    share = os.environ.get("GOF_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        dist.all_reduce(torch.zeros(1, device=dev))      # the communicator exists before a collective is issued from the autograd thread
        # one rank per GPU, or fail loudly: two ranks on one device would report a "scaling" that is time-slicing (and RCCL would hang or
        # crawl).  GOF_BENCH_SHARE_GPU=1 is the development override (gloo, fewer GPUs than ranks).
        ids = [None] * world
        try:
            me = str(torch.cuda.get_device_properties(dev).uuid)
        except Exception:
            me = "%s#%d" % (os.uname().nodename, dev_index)
        dist.all_gather_object(ids, me)
        if len(set(ids)) != world and not share:
            raise SystemExit("bench.py: %d ranks but only %d distinct GPUs (%s): one rank per GPU, or GOF_BENCH_SHARE_GPU=1 for a development run" % (world, len(set(ids)), ids))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    import synthetic_scenes as S
    from gpu_common import to_dev, settings_from
    from diff_gaussian_rasterization import GaussianRasterizer, _backend as B

    P, W, H = args.gaussians, args.width, args.height
    focal = 1200.0 * W / 1600.0
    # one scene (identical Gaussians on every rank = replica), one view per rank per step: weak scaling over views
    sc = S.scene_frustum(P, W=W, H=H, focal=focal, seed=0, kernel_size=args.kernel_size)
    sd = to_dev(sc, dev)
    params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    rast = GaussianRasterizer(settings_from(sd))
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    dL = torch.randn((9, H, W), generator=g).to(dev)

    if distributed:
        from dp import GradientAllReducer
        # the SH gradient travels as 12 B per Gaussian and view (all-gather + local expansion) instead of 192 B (dp/reducer.py);
        # GOF_DP_DENSE_SH=1 all-reduces the dense 236 B instead (A/B)
        reducer = GradientAllReducer(list(params.values()), sh_params=None if os.environ.get("GOF_DP_DENSE_SH") == "1" else [params["shs"]])

    ex_events = []                                # (start, end) torch events around the exchange of every timed step

    def step(timed=False):
        for p in params.values():
            p.grad = None
        means2D.grad = None
        color, radii = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"],
                            scales=params["scales"], rotations=params["rotations"])
        color.backward(dL)
        if distributed:
            if timed:                                 # the exchange ends with work.wait() on the current stream: events on it bracket the exposed part
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                reducer.all_reduce()
                e1.record()
                ex_events.append((e0, e1))
            else:
                reducer.all_reduce()
        return radii

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # ---- the timed region: exactly `steps` steps, no instrumentation inside (the exchange events are two stream markers) ----
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    exchange_ms = (sum(a.elapsed_time(b) for a, b in ex_events) / len(ex_events)) if ex_events else None
    # ---- per-kernel durations: a SEPARATE pass of instrumented steps (HIP events recorded by the library on its launch stream
    # around every launch; each event pair drains the queue between two kernels, so this pass is ~2 % slower than the timed one) ----
    kernel_times = None
    prof_steps = max(4, min(16, args.steps))
    if rank == 0:
        B.profile_enable(True)
    if distributed:
        reducer.profile = True                    # stream events around every phase of the exchange, on every rank
    for _ in range(prof_steps):
        step()
    fence()
    if rank == 0:
        kernel_times = B.profile_report()
        B.profile_enable(False)
    exchange_phases = None
    if distributed:
        reducer.profile = False
        mine = dict(reducer.breakdown(), rank=rank, buckets_in_place_and_packed=reducer.last_buckets)
        exchange_phases = [None] * world
        dist.all_gather_object(exchange_phases, mine)
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if exchange_ms is not None:
            t = torch.tensor([exchange_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exchange_ms = float(t.item())
        assert dist.get_world_size() == args.gpus

    ms_per_step = 1e3 * elapsed / args.steps
    iters_per_s = world * args.steps / elapsed        # views (= training iterations of the reference) per second, whole job

    out = None
    if rank == 0:
        # ---- per-stage timing with HIP events on the launch stream + roofline of the dominant HBM kernel ----
        stage = stage_times(B, sd, dL, dev, kernel_times, prof_steps)
        out = {
            "metric": "train iters/sec (fwd+bwd of the rasterizer, 1 view/GPU/step), S1M synthetic",
            "value": round(iters_per_s, 3), "unit": "iters/s", "n_gpus": (dist.get_world_size() if distributed else 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S1M: %d Gaussians @ %dx%d, SH degree 3, kernel_size %.2f, fwd+bwd%s" % (
                P, W, H, args.kernel_size, (" + RCCL gradient exchange (%s)" % reducer.last_exchange) if distributed else ""),
                "num_rendered": stage["R"], "fwd_Msplats_per_s": round(P / (stage["fwd_ms"] * 1e-3) / 1e6, 2),
                "fwd_ms": round(stage["fwd_ms"], 4), "bwd_ms": round(stage["bwd_ms"], 4),
                "fwd_overlapped_second_stream_ms": round(stage["fwd_overlapped_ms"], 4),      # preprocess_fwd's stage 2, beside the binning chain (not in fwd_ms)
                "parallelism": "dp%d (views)" % world},
            "roofline": stage["roofline"],
        }
        if distributed:
            n = world
            kind = reducer.last_exchange
            wire = (2.0 * (n - 1) / n * 236 * P) if kind == "dense" else (2.0 * (n - 1) / n * 44 * P + (n - 1) * 12.0 * (P + 1))
            out["exchange"] = {"kind": kind, "ms": round(exchange_ms, 4), "wire_bytes_per_rank": int(wire),
                               "wire_GBps_per_rank": round(wire / (exchange_ms * 1e-3) / 1e9, 1) if exchange_ms else None,
                               "backend": dist.get_backend(), "world_size": dist.get_world_size(),
                               # mean ms per step and rank of every phase (instrumented pass, not the timed one): pack = colour-gradient
                               # pack + camera row (inside the backward), gather_wait / reduce_wait = EXPOSED part of the all-gather /
                               # all-reduce (compute stream waiting for the communication stream), expand = SH expansion kernel,
                               # bucket_pack / bucket_unpack = copies of gradients outside the rasterizer's allocation (none here)
                               "per_rank_phases_ms": exchange_phases,
                               # optimistic pools on the data-parallel path (round 5): frames of rank 0 that had to be repeated because a
                               # pool sized from earlier frames was too small (forward again / blend stage again), and synchronising
                               # read-backs in front of a backward (the first frame of the shape only)
                               "pools_rank0": {k: B._stats[k] for k in ("mask_pool_redone_frames", "record_pool_redone_backwards", "backward_queries", "fused_redone_frames")},
                               "what": "exposed time of GradientAllReducer.all_reduce() per step (max over ranks), from stream events around it; "
                                       "the all-gather of the colour gradient starts inside the backward and overlaps preprocess_bwd"}
        if world == 1 and not args.no_full_loop:
            out["full_loop"] = full_loop(sd, dev, W, H)
        if world == 1 and not args.no_clustered and P == 1_000_000:
            out["clustered"] = clustered_leg(dev, P, W, H, focal, args.kernel_size, out["ms_per_step"])
        if world == 1 and not args.no_views and P == 1_000_000:
            out["views"] = views_leg(dev, sc, out["ms_per_step"])
        if world == 1 and not args.no_kernel_size_leg and P == 1_000_000 and args.kernel_size == 0.0:
            import synthetic_scenes as S_
            out["kernel_size_0p1"] = scene_leg(dev, S_.scene_frustum(P, W=W, H=H, focal=focal, seed=0, kernel_size=0.1), "S1M, kernel_size 0.1", out["ms_per_step"], steps=20, warmup=3)
        if world == 1 and not args.no_large_p:
            out["large_p"] = large_p_leg(dev, out["ms_per_step"])
        if world == 1 and not args.no_reference:
            out["reference_same_gpu"] = reference_leg(sd, dL, out["ms_per_step"])
        if world == 1 and not args.no_integrate:
            out["integrate"] = integrate_leg(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, W, H, focal)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def stage_times(B, sd, dL, dev, kernel_times, steps):
    """Per-kernel average durations (HIP events recorded by the library on its launch stream during the timed
    region) and the HBM roofline of the dominant kernel on ALGORITHMIC bytes."""
    from gpu_common import product_forward_raw, fetch
    res = product_forward_raw(sd)
    torch.cuda.synchronize()
    P, W, H, R = res["view"].P, sd["W"], sd["H"], int(res["R"])
    N = W * H
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = fetch(res, "ranges").view(np.uint32).reshape(-1, 2).astype(np.int64)
    lens = ranges[:, 1] - ranges[:, 0]
    last = np.zeros((gy * 16, gx * 16), np.int64)
    last[:H, :W] = fetch(res, "n_contrib").view(np.uint32).reshape(2, H, W)[0]
    tile_max_last = last.reshape(gy, 16, gx, 16).max(axis=(1, 3)).ravel()
    r_staged_bwd = int(tile_max_last.sum())                          # entries the backward has to look at
    r_visited_fwd = int(np.minimum(lens, tile_max_last + 1).sum())   # entries the forward has to look at
    p_visible = int((res["radii"] > 0).sum().item())
    tile_passes = (int(np.ceil(np.log2(max(2, gx * gy)))) + 1 + 7) // 8
    pairs = int(fetch(res, "contrib_pairs").astype(np.int64).sum())   # contributing (pixel, list entry) pairs of this view
    alg_bytes = {                                                    # SURVEY.md 8(d) per-unit figures x units, exactly as written there
        "preprocess_fwd": P * (236 + 119),
        "scan_tiles": 8 * P,
        "sort_gaussians_by_depth": 4 * 16 * P + 4 * P,               # 4 passes: 8 B read + 8 B written; one histogram read for all passes
        "emit_instances": 24 * P + 8 * R,
        "sort_instances_by_tile": tile_passes * (16 * R + 4 * R),     # per pass: 8 B read + 8 B written, + histogram read
        "tile_ranges": 8 * R + 8 * gx * gy,
        "blend_forward": 72 * r_visited_fwd + 60 * N,                # 8(d): 60 B state + 12 B colour per visited entry, 60 B per pixel
        "blend_backward": 72 * r_staged_bwd + 96 * N + 76 * p_visible,   # 8(d): 72 B per staged entry, 60 + 36 B per pixel, accumulators once
        "preprocess_bwd": p_visible * (316 + 232),
        "gather_tile_partials": 64 * r_staged_bwd + 4 * R + 8 * P + 4 * P + 68 * P,   # partial records + slot words + counts/offsets + blend weight read, 17 floats per Gaussian written
        "backward_memsets": 4 * R,                                         # the slot words (a record pool instead of a record per instance, round 4)
    }
    # bytes this design moves on top of 8(d)'s list: the contributor masks (1 bit per pixel and visited entry = 32 B per entry),
    # written by the forward, read by the backward, and the 32 B footprint conic per entry the forward's cull scan reads
    extra_bytes = {"blend_forward": (32 + 32) * r_visited_fwd, "blend_backward": (32 + 68) * r_staged_bwd}   # masks read; partial record + slot word written
    if "preprocess_fwd_heavy" in kernel_times:
        # round 5: the sync-free forward runs the per-Gaussian kernel in two stages -- "preprocess_fwd" = the culls + what binning reads
        # (40 B read, 24 B written per Gaussian), "preprocess_fwd_heavy" = the rest (8(d)'s 355 B less those 24) on a second stream
        # BESIDE the binning chain: its time is not part of the forward's critical path (fwd_ms below leaves it out)
        alg_bytes["preprocess_fwd"] = P * 64
        alg_bytes["preprocess_fwd_heavy"] = P * (236 + 119 - 24)
    kernels = {}
    for name, rec in kernel_times.items():
        avg_ms = rec["total_ms"] / max(1, rec["calls"])
        ent = {"avg_ms": round(avg_ms, 5), "calls": rec["calls"]}
        if name in alg_bytes and avg_ms > 0:
            ent["alg_MB"] = round(alg_bytes[name] / 1e6, 2)
            ent["GBps"] = round(alg_bytes[name] / (avg_ms * 1e-3) / 1e9, 1)
            if name in extra_bytes:
                ent["design_extra_MB"] = round(extra_bytes[name] / 1e6, 2)
        kernels[name] = ent
    fwd_names = ("preprocess_fwd", "sort_gaussians_by_depth", "scan_tiles", "emit_instances", "sort_instances_by_tile", "tile_ranges", "order_tiles", "blend_forward", "order_tiles_bw")
    fwd_ms = sum(kernels[k]["avg_ms"] for k in fwd_names if k in kernels)
    bwd_ms = sum(kernels[k]["avg_ms"] for k in ("backward_memsets", "blend_backward", "gather_tile_partials", "preprocess_bwd") if k in kernels)
    dom = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["calls"])
    d = kernels[dom]
    achieved = d.get("GBps", 0.0)
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE): rocprofv3 cannot run
    # inside this process, so the figure is READ FROM the committed counter pass of the same workload (profiles/), labelled with
    # its source and with the hash of the kernel sources it was collected on; null if that hash is not the current one
    traffic = valu_issue_frac = pmc_source = None
    pmc_file = os.path.join(ROOT, "profiles", PMC_FILE)
    if P == 1_000_000 and (W, H) == (1600, 1063) and os.path.exists(pmc_file):
        pmc_all = json.load(open(pmc_file))
        pmc_source = {"file": "profiles/" + PMC_FILE, "sha16_by_kernel_at_collection": pmc_all.get("_sha16_by_kernel"),
                      "collected_by": "rocprofv3 --pmc passes of tests/devtools/dev_pmc.py (separate FETCH_SIZE / WRITE_SIZE / SQ passes)"}
        pmc_source["current_sha16_of_the_dominant_kernel"] = kernel_sha16(dom)
        stale = []
        for name, ent in kernels.items():               # every kernel's counter traffic, but only from a pass on ITS current source
            if name in pmc_all and isinstance(pmc_all[name], dict) and "hbm_bytes_corrected" in pmc_all[name]:
                if pmc_pass_is_current(pmc_all, name):
                    ent["traffic_MB"] = round(pmc_all[name]["hbm_bytes_corrected"] / 1e6, 1)
                else:
                    ent["traffic_MB"] = None
                    stale.append(name)
        pmc_source["kernels_whose_source_changed_since_the_pass"] = stale
        if pmc_pass_is_current(pmc_all, dom):
            traffic = pmc_all.get(dom, {}).get("hbm_bytes_corrected")
            valu_issue_frac = pmc_all.get(dom, {}).get("valu_issue_frac")
    # what actually bounds the two blend kernels: vector-ALU work.  "Useful" flop per contributing pair = the arithmetic the
    # reference's source spends on a contributing (pixel, Gaussian) pair (forward 85, backward 190; FMA = 2);
    # peak = 157.3 TFLOP/s fp32 vector (MI355X_MICROARCH.md)
    valu = {}
    for name, flop in (("blend_forward", FLOP_PER_PAIR_FWD), ("blend_backward", FLOP_PER_PAIR_BWD)):
        if name in kernels and kernels[name]["avg_ms"] > 0:
            tf = flop * pairs / (kernels[name]["avg_ms"] * 1e-3) / 1e12
            valu[name] = {"bound": "valu", "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
                          "flop_per_pair": flop, "pairs": pairs, "avg_ms": kernels[name]["avg_ms"]}
    roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": pmc_source,
            "avg_ms": d["avg_ms"], "alg_bytes": alg_bytes.get(dom), "design_extra_bytes": extra_bytes.get(dom),
            "valu_issue_frac": None if valu_issue_frac is None else round(valu_issue_frac, 3),
            "valu": valu,
            "note": "dominant kernel by time on SURVEY 8(d)'s algorithmic bytes (its HBM floor); the two blend kernels are bound by "
                    "vector-ALU issue, not by HBM -- their roofline is under 'valu'; the HBM-bound stages are listed under 'kernels'",
            "kernels": kernels,
            "workload": {"R": R, "P_visible": p_visible, "R_visited_fwd": r_visited_fwd, "R_staged_bwd": r_staged_bwd, "contributing_pairs": pairs,
                         "mean_tile_list": round(float(lens.mean()), 1), "mean_last_contributor": round(float(last[:H, :W].mean()), 1)},
            "workspace": workspace_report(B, P, W, H, R, r_staged_bwd)}
    # stage 2 of the per-Gaussian kernel runs on the library's second stream BESIDE the binning chain: GPU work that is not on the forward's
    # critical path (fwd_ms leaves it out) but competes with it for the CUs -- reported next to it, labelled
    heavy_ms = kernels["preprocess_fwd_heavy"]["avg_ms"] if "preprocess_fwd_heavy" in kernels else 0.0
    return {"fwd_ms": fwd_ms, "bwd_ms": bwd_ms, "R": R, "roofline": roof, "fwd_overlapped_ms": heavy_ms}


def workspace_report(B, P, W, H, R, staged):
    """Bytes of the caller-owned workspaces of one forward + backward, and what an INSTANCE (one (tile, Gaussian) pair of the sorted
    list) costs.  Binning workspace: sort state 16 B + contributor masks 32 B.  Backward scratch (round 4): a slot word per instance
    + a 64-byte partial gradient record per STAGED instance (the pool is sized from gof_backward_query: `staged` of R) -- rounds 2-3
    held a record per instance, 69 B.  The reference's BinningState holds 24 B (rasterizer_impl.h:60-70): a stated deviation
    (DESIGN.md section 7): masks and records buy the backward without re-derived decisions and without atomics."""
    lib = B.lib
    geom, image = int(lib.gof_geom_bytes_forward(P)), int(lib.gof_image_bytes(W, H))      # (what a forward / backward pair allocates: without the query's 16 B per Gaussian)
    binning_full = int(lib.gof_binning_bytes(R, W, H))
    # what the shipped binding allocates in steady state: the instance capacity (1.25 x the count + 64 Ki) and a mask pool 1.25 x the
    # largest request seen (learnt at the first backward) on top of the sort state
    key = [k for k in B._capacity if k[1:] == (P, W, H)]
    cap = B._capacity[key[0]] if key else R
    sub = B._mask_pool_subchunks(key[0]) if key else None
    binning = int(lib.gof_binning_bytes(cap, W, H) if sub is None else lib.gof_binning_bytes_for(cap, W, H, sub))
    scratch_full = int(lib.gof_backward_scratch_bytes(P, R))
    # (round 5: the same optimistic pools when a data-parallel reducer starts its exchange inside the backward -- the blend stage is
    # verified against the forward's counters before the colour gradient is handed over, _backend.rasterize_gaussians_backward)
    scratch = int(lib.gof_backward_scratch_bytes_for(P, cap, min(cap, int(max(B._staged_need.get(key[0], staged) if key else staged, staged) * 1.25) + 4096)))
    d_bin = binning / max(R, 1)
    fixed = int(lib.gof_backward_scratch_bytes_for(P, 0, 0))
    d_scr = (scratch - fixed) / max(R, 1)
    return {"geometry_bytes": geom, "image_bytes": image, "binning_bytes": binning, "binning_bytes_worst_case": binning_full,
            "mask_subchunks_requested": (B._mask_need.get(key[0]) if key else None), "mask_subchunks_held": sub, "instance_capacity": cap,
            "backward_scratch_bytes": scratch,
            "backward_scratch_bytes_worst_case": scratch_full, "staged_fraction_of_instances": round(staged / max(R, 1), 4),
            "per_gaussian_geometry_bytes": round(geom / max(P, 1), 1), "per_instance_binning_bytes": round(d_bin, 1),
            "per_instance_backward_scratch_bytes": round(d_scr, 1), "per_instance_total_bytes": round(d_bin + d_scr, 1),
            "reference_per_instance_bytes": 24, "reference_per_gaussian_bytes": 119}


def clustered_leg(dev, P, W, H, focal, kernel_size, s1m_ms, steps=20, warmup=3):
    """The same step on a HEAVY-TAILED scene (synthetic_scenes.scene_clustered, "S1M-clustered": 70 % of the Gaussians in five
    semi-transparent blobs, 2 % large splats at the back, sparse background -- tile lists of 1300 ... 14500 entries of which 220 ...
    2060 are walked, where S1M's tiles all cost the same): what the tile scheduler (order_tiles + pop_tile: tiles ranked by cost,
    dealt to the XCD queues, heaviest first) is for.  An extra key next to the headline, not instead of it.  The scheduler's
    efficiency against the work-proportional ideal (per-tile clocks of an instrumented build) and the A/B against round 2's
    static map are in profiles/r03_tile_schedule.md."""
    import synthetic_scenes as S
    from gpu_common import to_dev, settings_from, product_forward_raw, fetch
    from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
    sc = S.scene_clustered(P, W=W, H=H, focal=focal, seed=0, kernel_size=kernel_size)
    sd = to_dev(sc, dev)
    params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    rast = GaussianRasterizer(settings_from(sd))
    dL = torch.randn((9, H, W), generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)

    def step():
        for p in params.values():
            p.grad = None
        means2D.grad = None
        color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"],
                        scales=params["scales"], rotations=params["rotations"])
        color.backward(dL)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    B.profile_enable(True)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    rep = B.profile_report()
    B.profile_enable(False)
    res = product_forward_raw(sd)
    rg = fetch(res, "ranges").view(np.uint32).reshape(-1, 2).astype(np.int64)
    lens = rg[:, 1] - rg[:, 0]
    walked = fetch(res, "tile_cost").astype(np.int64)
    pct = lambda a: [int(x) for x in np.percentile(a, [0, 50, 90, 99, 100])]   # noqa: E731
    return {"workload": "S1M-clustered: %d Gaussians @ %dx%d, SH degree 3, kernel_size %.2f, fwd+bwd" % (P, W, H, kernel_size),
            "ms_per_step": round(ms, 4), "iters_per_s": round(1e3 / ms, 2), "steps": steps, "num_rendered": int(res["R"]),
            "tile_list_length_pct_0_50_90_99_100": pct(lens), "entries_walked_per_tile_pct_0_50_90_99_100": pct(walked),
            "vs_s1m_ms_per_step": round(ms / s1m_ms, 3),
            "kernels_ms": {k: round(v["total_ms"] / max(1, v["calls"]), 4) for k, v in rep.items()}}


def scene_leg(dev, sc, label, s1m_ms, steps=10, warmup=3, with_full_loop=False):
    """fwd + bwd of one view of another synthetic scene through the autograd surface, OUTSIDE the headline's timed region: ms per step,
    per-stage durations (the library's HIP events) with GB/s on SURVEY 8(d)'s algorithmic bytes, optionally the full training
    iteration.  Used for `kernel_size_0p1` (BASELINE config 2 names kernel_size in {0.0, 0.1}: the 2D low-pass filter of
    forward.cu:112-118 on the headline scene) and `large_p` (round 5: real captures hold millions of small Gaussians -- there
    preprocess / SH / Adam traffic is the step, not the blends)."""
    from gpu_common import to_dev, settings_from
    from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
    W, H, P = sc["W"], sc["H"], int(sc["means3D"].shape[0])
    sd = to_dev(sc, dev)
    params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    rast = GaussianRasterizer(settings_from(sd))
    dL = torch.randn((9, H, W), generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)

    def step():
        for p in params.values():
            p.grad = None
        means2D.grad = None
        color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"],
                        scales=params["scales"], rotations=params["rotations"])
        color.backward(dL)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    B.profile_enable(True)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    rep = B.profile_report()
    B.profile_enable(False)
    st = stage_times(B, sd, dL, dev, rep, 4)
    kern = {k: {kk: v[kk] for kk in ("avg_ms", "alg_MB", "GBps") if kk in v} for k, v in st["roofline"]["kernels"].items()}
    for v in kern.values():
        if "GBps" in v:
            v["frac_of_8TBps"] = round(v["GBps"] / 8000.0, 3)
    out = {"workload": "%s: %d Gaussians @ %dx%d, SH degree 3, kernel_size %.2f, fwd+bwd" % (label, P, W, H, sc["kernel_size"]),
           "ms_per_step": round(ms, 4), "iters_per_s": round(1e3 / ms, 2), "steps": steps, "num_rendered": st["R"],
           "fwd_ms": round(st["fwd_ms"], 4), "bwd_ms": round(st["bwd_ms"], 4), "vs_s1m_ms_per_step": round(ms / s1m_ms, 3),
           "workload_counts": st["roofline"]["workload"], "kernels": kern}
    del params, means2D, rast
    if with_full_loop:
        fl = full_loop(sd, dev, W, H, steps=5, warmup=2)
        out["full_loop"] = {k: fl[k] for k in ("ms_per_iter", "iters_per_s", "epilogue_kernels_ms", "adam_GBps") if k in fl}
        out["full_loop"]["launcher_default_ms_per_iter"] = fl["launcher_default"]["ms_per_iter"]
        out["full_loop"]["one_call_loss_split_sh_ms_per_iter"] = fl["one_call_loss_split_sh"]["ms_per_iter"]
    del sd
    torch.cuda.empty_cache()
    return out


def large_p_leg(dev, s1m_ms):
    """Round 5: the regime of real scenes.  `bicycle_like`: 6M Gaussians at 1237x822 (Mip-NeRF360 bicycle, images_4: the resolution
    of BASELINE config 3), median projected sigma 1.5 px; `s5m`: 5M Gaussians of the same size at the headline resolution."""
    import synthetic_scenes as S
    out = {}
    for label, mk in (("bicycle_like_6M", lambda: S.scene_frustum(6_000_000, W=1237, H=822, focal=1237.0 * 0.75, seed=0, sigma_px=1.5)),
                      ("s5m", lambda: S.scene_frustum(5_000_000, seed=0, sigma_px=1.5))):
        try:
            out[label] = scene_leg(dev, mk(), label, s1m_ms, steps=10, warmup=3, with_full_loop=True)
        except Exception as e:      # an extra leg must never take the headline down
            out[label] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def views_leg(dev, sc, s1m_ms, n_views=8, steps=40, warmup=8):
    """A training run renders a DIFFERENT camera every iteration (train.py:100-104): instance count (the sync-free forward's learnt
    capacity), tile costs and their order, L2 / MALL contents all change from step to step, where the headline renders one view over
    and over.  This leg cycles `n_views` posed cameras (synthetic_scenes.other_view: the same cloud seen from rigidly moved cameras)
    inside its timed region: ms per step over the cycle, the per-view spread, and how often the fused forward had to redo a frame
    because the instance count outgrew the capacity learnt from the previous views.  Reported beside the headline, not as it."""
    import synthetic_scenes as S
    from gpu_common import to_dev, settings_from
    from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
    views = [to_dev(sc if v == 0 else S.other_view(sc, v), dev) for v in range(n_views)]
    base = views[0]
    params = {k: base[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    rasts = [GaussianRasterizer(settings_from(v)) for v in views]
    H, W = base["H"], base["W"]
    dL = torch.randn((9, H, W), generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)
    counts = []

    def step(i):
        for p in params.values():
            p.grad = None
        means2D.grad = None
        color, _ = rasts[i % n_views](means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"],
                                      scales=params["scales"], rotations=params["rotations"])
        color.backward(dL)
    redo0 = B._stats["fused_redone_frames"]
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    redo = B._stats["fused_redone_frames"] - redo0
    per_view = []
    for v in range(n_views):                       # each view on its own (the headline's pattern), for the spread
        for _ in range(2):
            step(v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step(v)
        torch.cuda.synchronize()
        per_view.append(1e3 * (time.perf_counter() - t0) / 5)
        counts.append(int(B._stats["last_num_rendered"]))
    return {"workload": "S1M seen from %d posed cameras, one per step, cycled inside the timed region, fwd+bwd" % n_views,
            "ms_per_step": round(ms, 4), "iters_per_s": round(1e3 / ms, 2), "steps": steps, "vs_s1m_ms_per_step": round(ms / s1m_ms, 3),
            "per_view_alone_ms": [round(x, 4) for x in per_view], "num_rendered_per_view": counts,
            "mean_of_the_views_alone_ms": round(float(np.mean(per_view)), 4),
            "fused_forward_redone_frames": redo}


def reference_leg(sd, dL, product_ms):
    """The reference's OWN kernels on this GPU: submodules/diff-gaussian-rasterization/cuda_rasterizer/*.cu compiled for gfx950 where
    they lie by oracle/build_ref.sh (default FMA contraction, as a user's build would be) -- oracle/_ref/libgof_cudaref.so, TEST
    INFRASTRUCTURE, timed here OUTSIDE the timed region of the headline, on the same scene, everything resident on the device.
    BASELINE.md holds no published number for this metric (vs_baseline stays null); this is the same-GPU yardstick."""
    try:
        import reference_binding as rb
        if not rb.available(""):
            return {"available": False, "why": "oracle/_ref/libgof_cudaref.so is not in this checkout (built where /root/reference exists)"}
        ref = rb.Reference(sd, "")
        fwd_ms, bwd_ms = ref.time_forward_backward(dL.contiguous())
        ms = fwd_ms + bwd_ms
        return {"available": True, "kind": "the reference's CUDA sources compiled for gfx950 (oracle/build_ref.sh), default FMA contraction",
                "fwd_ms": round(fwd_ms, 3), "bwd_ms": round(bwd_ms, 3), "ms_per_step": round(ms, 3), "iters_per_s": round(1e3 / ms, 2),
                "num_rendered": int(ref.R), "product_speedup": round(ms / product_ms, 2),
                "note": "3 iterations after 1 warm-up, medians; host-side wall clock around the two synchronising wrapper calls"}
    except Exception as e:      # the yardstick must never take the headline down
        return {"available": False, "why": "%s: %s" % (type(e).__name__, e)}


def full_loop(sd, dev, W, H, steps=10, warmup=3, only_inline=False, only_launcher=False):
    """SURVEY.md 8(d)(i) "full-loop variant": one complete training iteration of train.py:125-190, 263-265 on the same
    scene -- the parameter activations render() reads (scene/gaussian_model.py:157-194, HIP), rasterizer forward, the reference's loss
    (L1 + D-SSIM + depth-normal consistency + distortion), backward, Adam over the 59 floats per Gaussian -- with the
    HIP training epilogue (train_epilogue/: ssim, depth_to_normal, FusedAdam).  Reported beside the headline, not as it."""
    import math
    import types
    import train_epilogue as T
    from gpu_common import settings_from
    from diff_gaussian_rasterization import GaussianRasterizer, SplitSH, _backend as B
    raw = {
        "xyz": sd["means3D"].clone(), "f_dc": sd["shs"][:, :1].clone(), "f_rest": sd["shs"][:, 1:].clone(),
        "opacity": torch.logit(sd["opacities"].clamp(1e-4, 1 - 1e-4)), "scaling": torch.log(sd["scales"]), "rotation": sd["rotations"].clone(),
    }
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3}
    params, opt = {}, None

    def reset():
        """Every leg starts from the same parameters and a fresh optimizer state (the legs would otherwise time different scenes:
        ~13 Adam steps against a random target image change the workload)."""
        nonlocal opt
        params.clear()
        params.update({k: torch.nn.Parameter(v.clone().contiguous()) for k, v in raw.items()})
        opt = T.FusedAdam([{"params": [p], "lr": lrs[k], "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)   # gaussian_model.py:349-360
    reset()
    rast = GaussianRasterizer(settings_from(sd))
    means2D = torch.zeros_like(params["xyz"], requires_grad=True)
    gt = torch.rand((3, H, W), generator=torch.Generator().manual_seed(7)).to(dev)
    # (the launcher hands train.py a Camera whose world_view_transform remembers its inverse: train_epilogue/pose.py)
    view = types.SimpleNamespace(world_view_transform=T.PoseMatrix.wrap(sd["viewmatrix"]), image_width=W, image_height=H,
                                 FoVx=2 * math.atan(sd["tanfovx"]), FoVy=2 * math.atan(sd["tanfovy"]))
    lambda_dssim, lambda_dn, lambda_dist = 0.2, 0.05, 100.0        # arguments/__init__.py defaults
    filter_3D = (sd["scales"].min(dim=1, keepdim=True).values * 0.1).contiguous()           # a small 3D smoothing filter (compute_3D_filter's role)

    from train_epilogue import deferred as Dl

    def iteration(one_call_loss=False, split_sh=False, launcher=False):
        # launcher: the helper names as launch/run_reference_script.py binds them -- deferred evaluation on (train_epilogue/deferred.py):
        # the SAME lines below then cost no launch until loss.backward(), which is one fused call
        # get_features (gaussian_model.py:173-176): the concatenation as the reference forms it, or -- with the launcher's default
        # rebinding -- the two stored tensors handed over as they are (SplitSH)
        shs = SplitSH(params["f_dc"], params["f_rest"]) if split_sh else torch.cat((params["f_dc"], params["f_rest"]), dim=1)
        A = T.activations                                                                    # gaussian_renderer/__init__.py:60,70-71
        rendering, radii = rast(means3D=params["xyz"], means2D=means2D, shs=shs,
                                opacities=A.opacity_with_3D_filter(params["opacity"], params["scaling"], filter_3D),
                                scales=A.scaling_with_3D_filter(params["scaling"], filter_3D), rotations=A.rotation(params["rotation"]))
        if one_call_loss:                                                                    # train.py:150-188 as ONE operator (gof_train_loss)
            loss = T.training_loss(rendering, gt, view, lambda_dssim, lambda_dn, lambda_dist).loss
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss
        l1_loss, ssim, depth_to_normal = (Dl.l1_loss, Dl.ssim, Dl.depth_to_normal) if launcher else (T.l1_loss, T.ssim, T.depth_to_normal)
        image = rendering[:3]
        rgb_loss = (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))       # train.py:156-161
        distortion_loss = rendering[8].mean()                                                # :164-167
        depth_normal = depth_to_normal(view, rendering[6][None])[0].permute(2, 0, 1)         # :170-172
        render_normal = torch.nn.functional.normalize(rendering[3:6], p=2, dim=0)            # :174-175
        c2w = (view.world_view_transform.T).inverse()                                        # :177
        world = (c2w[:3, :3] @ render_normal.reshape(3, -1)).reshape(3, H, W)                # :178-179
        depth_normal_loss = (1 - (world * depth_normal).sum(dim=0)).mean()                   # :181-182
        loss = rgb_loss + depth_normal_loss * lambda_dn + distortion_loss * lambda_dist      # :188
        loss.backward()                                                                      # :189
        opt.step()                                                                           # :264
        opt.zero_grad(set_to_none=True)                                                      # :265
        return loss

    if only_launcher:          # (tests/devtools/dev_full_loop_trace.py launcher: a kernel trace of what the launcher runs by default)
        Dl.enable(True)
        try:
            for _ in range(warmup):
                iteration(False, True, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                iteration(False, True, True)
            torch.cuda.synchronize()
        finally:
            Dl.enable(False)
        ms = 1e3 * (time.perf_counter() - t0) / steps
        return {"ms_per_iter": round(ms, 4), "iters_per_s": round(1e3 / ms, 2), "steps": steps, "deferred_loss": dict(Dl.stats)}
    for _ in range(warmup):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    if only_inline:            # (tests/devtools/dev_full_loop_trace.py: a kernel trace of the unchanged train.py composition alone)
        return {"ms_per_iter": round(ms, 4), "iters_per_s": round(1e3 / ms, 2), "steps": steps}
    reset()
    for _ in range(warmup):
        iteration(False, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration(False, True)
    torch.cuda.synchronize()
    ms_eager = 1e3 * (time.perf_counter() - t0) / steps
    reset()
    Dl.enable(True)
    fused0 = Dl.stats["fused_backwards"]
    try:
        for _ in range(warmup):
            iteration(False, True, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            iteration(False, True, True)
        torch.cuda.synchronize()
        ms_launcher = 1e3 * (time.perf_counter() - t0) / steps
    finally:
        Dl.enable(False)
    fused_calls = Dl.stats["fused_backwards"] - fused0
    reset()
    for _ in range(warmup):
        iteration(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration(True)
    torch.cuda.synchronize()
    ms_one = 1e3 * (time.perf_counter() - t0) / steps
    reset()
    for _ in range(warmup):
        iteration(True, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration(True, True)
    torch.cuda.synchronize()
    ms_split = 1e3 * (time.perf_counter() - t0) / steps
    reset()
    B.profile_enable(True)                    # epilogue kernel durations: three more iterations with the library's HIP events
    for _ in range(3):
        iteration()
    kt = B.profile_report()
    B.profile_enable(False)
    ep = {k: round(v["total_ms"] / max(1, v["calls"]), 5) for k, v in kt.items()
          if k in ("ssim_forward", "ssim_backward", "depth_to_normal", "depth_to_normal_backward", "adam_step", "act_scaling", "act_opacity",
                   "act_rotation", "act_scaling_backward", "act_opacity_backward", "act_rotation_backward")}
    n_floats = sum(p.numel() for p in params.values())
    out = {"ms_per_iter": round(ms, 4), "iters_per_s": round(1e3 / ms, 2), "steps": steps,
           "includes": "3D-filter activations + rasterizer fwd/bwd + L1/D-SSIM/depth-normal/distortion loss + Adam (59 floats/Gaussian); "
                       "the camera pose is the launcher's PoseMatrix (train.py:177-179: inverse computed once, 3x3 block applied by one streaming launch)",
           "epilogue_kernels_ms": ep,
           "launcher_default": {"ms_per_iter": round(ms_launcher, 4), "iters_per_s": round(1e3 / ms_launcher, 2),
                                "losses_evaluated_by_one_fused_call": "%d of %d" % (fused_calls, steps + warmup),
                                "what": "the unchanged train.py as launch/run_reference_script.py runs it: the inline loss composition above with "
                                        "the helper names bound to train_epilogue/deferred.py (the script's lines collect coefficients, "
                                        "loss.backward() is one gof_train_loss call), GaussianModel.get_features rebound to the two stored SH "
                                        "tensors (SplitSH: no 192 B/Gaussian concatenation and gradient split per iteration)"},
           "launcher_eager_loss": {"ms_per_iter": round(ms_eager, 4), "iters_per_s": round(1e3 / ms_eager, 2),
                                   "what": "the same with GOF_EAGER_LOSS=1: every helper eager (one HIP launch pair each, torch for the rest) -- the "
                                           "launcher's default until round 6"},
           "one_call_loss": {"ms_per_iter": round(ms_one, 4), "iters_per_s": round(1e3 / ms_one, 2),
                             "what": "the same iteration with train.py:150-188 evaluated by train_epilogue.training_loss (gof_train_loss, "
                                     "five launches) instead of the inline torch composition; needs the 7-line train.py change of INTEGRATION.md"},
           "one_call_loss_split_sh": {"ms_per_iter": round(ms_split, 4), "iters_per_s": round(1e3 / ms_split, 2),
                                      "what": "... and the SH coefficients read from _features_dc / _features_rest directly (SplitSH, the launcher's "
                                              "default rebinding of GaussianModel.get_features) instead of their per-iteration concatenation"}}
    if ep.get("adam_step"):
        out["adam_GBps"] = round(28.0 * n_floats / (ep["adam_step"] * 1e-3) / 1e9, 1)     # p,g,m,v read + p,m,v written
    return out


def integrate_leg(dev, P=5_000_000, W=1600, H=1063):
    """BASELINE config 5 shape -- the opacity-field level-set query of extract_mesh.py: 5M Gaussians (sigma_px 1.5), 9 query points per
    Gaussian as GaussianModel.get_tetra_points builds them (45M points), one view at 1600x1063; the first call of a view (Gaussian
    side + pixel pass + point pass) and a later call of the same view (per-view cache: point binning + point pass only -- the
    pattern of the 8 bisection steps); then marching tetrahedra on a Freudenthal grid (CGAL Delaunay stand-in).  Per-kernel
    durations from the library's HIP events; HBM figures on the algorithmic bytes of DESIGN.md section 3."""
    import synthetic_scenes as S
    import diff_gaussian_rasterization as DGR
    from gpu_common import to_dev, settings_from
    from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
    sc = S.scene_frustum(P, W=W, H=H, seed=0, sigma_px=1.5)
    pts = torch.from_numpy(S.tetra_points(sc)).to(dev)
    sd = to_dev(sc, dev)
    r = GaussianRasterizer(settings_from(sd))

    def call():
        with DGR.integrate_view_key(("bench", 0)):
            return r.integrate(points3D=pts, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"],
                               scales=sd["scales"], rotations=sd["rotations"])
    call()
    torch.cuda.synchronize()
    legs = {}
    for name, clear in (("first_call_of_a_view", True), ("later_call_of_the_view", False)):
        ts, reps = [], None
        for _ in range(3):
            if clear:
                DGR.integrate_view_cache().clear()
            B.profile_enable(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = call()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
            reps = B.profile_report()
            B.profile_enable(False)
        legs[name] = {"wall_ms": round(min(ts), 3), "kernels_ms": {k: round(v["total_ms"] / max(1, v["calls"]), 4) for k, v in reps.items()}}
    n_in_view = int((out[1] != 1.0).sum().item())
    # one step of extract_mesh.py's view loop (:24-29) for a cached view: the call + the reduction over views, as the script composes it
    # (ones / zeros outputs, torch.where, torch.min) and with the reduction fused into the point pass (integrate_min_into)
    PNv = int(pts.shape[0])
    fa = torch.ones(PNv, device=dev); fc = torch.ones(PNv, 3, device=dev)
    def step_torch():
        nonlocal fa, fc
        o = call()
        fc = torch.where((o[1] < fa).reshape(-1, 1), o[2], fc)
        fa = torch.min(fa, o[1])
    def step_fused():
        with DGR.integrate_min_into(fa, fc):
            call()
    for name, fn in (("script_composition", step_torch), ("fused_into_the_point_pass", step_fused)):
        ts = []
        for _ in range(3):
            fa.fill_(1.0); fc.fill_(1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        legs.setdefault("view_loop_step_cached_view_with_colour", {})[name + "_wall_ms"] = round(min(ts), 3)
    del fa, fc
    DGR.integrate_view_cache().clear()
    PN = int(pts.shape[0])
    res = {"workload": "integrate: %d Gaussians, %d query points (9 per Gaussian), %dx%d, one view" % (P, PN, W, H),
           "points_evaluated": n_in_view, "Mpoints_per_s_cached": round(PN / (legs["later_call_of_the_view"]["wall_ms"] * 1e-3) / 1e6, 1), **legs}
    del pts, sd, out
    # marching tetrahedra on a 6-tets-per-cube grid with a sphere-like field: 24.6M tets
    import tetmesh
    n = 160
    verts, tets = S.freudenthal_tets(n, n, n)
    v = torch.from_numpy(verts).to(dev)
    t = torch.from_numpy(tets).to(dev)
    c = (v / n - 0.5)
    sdf = (c.norm(dim=1) - 0.35 + 0.03 * torch.sin(20 * c[:, 0]) * torch.cos(17 * c[:, 1]))[None]
    scl = torch.ones((1, v.shape[0], 1), device=dev)
    tetmesh.marching_tetrahedra(v[None], t, sdf, scl)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        m = tetmesh.marching_tetrahedra(v[None], t, sdf, scl)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    res["marching_tetrahedra"] = {"tets": int(t.shape[0]), "vertices": int(v.shape[0]), "faces": int(m[2][0].shape[0]), "wall_ms": round(min(ts), 3)}
    return res


def cpu_baseline(args, W, H, focal):
    """Oracle (CPU restatement of the reference, OpenMP) on a bounded sample of the same workload."""
    import oracle_binding as ob
    import synthetic_scenes as S
    Pc = min(args.cpu_baseline_gaussians, args.gaussians)
    sc = S.scene_frustum(Pc, W=W, H=H, focal=focal, seed=0, kernel_size=args.kernel_size)
    o = ob.OracleScene(sc)
    t0 = time.perf_counter()
    out, radii = o.forward()
    t1 = time.perf_counter()
    dL = np.random.default_rng(1).normal(size=out.shape).astype(np.float32)
    o.backward(dL)
    t2 = time.perf_counter()
    cores = int(ob.lib().gofref_num_threads())
    return {"value": round(1.0 / (t2 - t0), 4), "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": "oracle (OpenMP CPU restatement of the reference) fwd+bwd on %d Gaussians (%.0f%% of the workload's count) at %dx%d, one "
                      "iteration on the GPU box's host threads; fwd %.2fs bwd %.2fs" % (Pc, 100.0 * Pc / args.gaussians, W, H, t1 - t0, t2 - t1),
            "torch_cpu": torch_cpu_baseline()}


def torch_cpu_baseline():
    """north_star: "next to the reference's PyTorch-CPU render path timed on the same box's host cores" -- BASELINE config 1 (lego-like
    10 000 Gaussians @ 400x400, forward render).  The reference itself has no CPU path (gaussian_renderer/__init__.py:26 hard-wires
    "cuda"); its CPU-runnable form is the dense PyTorch restatement oracle/render_torch_cpu.py (test infrastructure, float64 and --
    the reference's precision -- float32), timed here beside the HIP forward of the same scene."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import render_torch_cpu as RT
    import synthetic_scenes as S
    from gpu_common import to_dev, product_forward_raw
    sc = S.scene_lego_like(10_000, 400, 400, seed=0)
    # dense whole-image tensor ops: beyond ~16 threads torch's intra-op pool only adds contention (measured on the 128-thread GPU
    # box: 43 s with 128 threads; 7.5 s with 8 threads in the build container), so the pool is bounded -- and restored afterwards
    n_before = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = {"workload": "BASELINE config 1: lego-like 10000 Gaussians @ 400x400, SH degree 3, forward render", "threads": torch.get_num_threads(),
           "host_cores": os.cpu_count()}
    try:
        t0 = time.perf_counter()
        RT.render_torch_cpu(sc, dtype=torch.float32)          # the reference's precision (float64 is the oracle-pin variant, tests/test_oracle_pins.py)
        out["torch_cpu_float32_s"] = round(time.perf_counter() - t0, 3)
    finally:
        torch.set_num_threads(n_before)
    sd = to_dev(sc)
    for _ in range(3):
        product_forward_raw(sd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        product_forward_raw(sd)
    torch.cuda.synchronize()
    out["hip_forward_ms"] = round(1e3 * (time.perf_counter() - t0) / 20, 4)
    out["renders_per_s"] = {"torch_cpu_float32": round(1.0 / out["torch_cpu_float32_s"], 3), "hip": round(1e3 / out["hip_forward_ms"], 1)}
    return out


if __name__ == "__main__":
    main()
