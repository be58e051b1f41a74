#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json: train iters/s (forward + backward of the
rasterizer, one view per step) on the synthetic S1M scene (1M Gaussians, 1600x1063, SH degree 3),
N GPUs data-parallel over views with an RCCL all-reduce of the Gaussian parameter gradients.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1063)
    ap.add_argument("--kernel-size", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-gaussians", type=int, default=100_000)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

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
        reducer = GradientAllReducer(list(params.values()))

    def step():
        for p in params.values():
            p.grad = None
        means2D.grad = None
        color, radii = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"],
                            scales=params["scales"], rotations=params["rotations"])
        color.backward(dL)
        if distributed:
            reducer.all_reduce()
        return radii

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / args.steps
    iters_per_s = world * args.steps / elapsed        # views (= training iterations of the reference) per second, whole job

    out = None
    if rank == 0:
        # ---- per-stage timing with HIP events on the launch stream + roofline of the dominant HBM kernel ----
        stage = stage_times(B, sd, dL, dev, reps=max(5, min(20, args.steps)))
        out = {
            "metric": "train iters/sec (fwd+bwd of the rasterizer, 1 view/GPU/step), S1M synthetic",
            "value": round(iters_per_s, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S1M: %d Gaussians @ %dx%d, SH degree 3, kernel_size %.2f, fwd+bwd%s" % (
                P, W, H, args.kernel_size, " + RCCL grad all-reduce (236 B/Gaussian)" if distributed else ""),
                "num_rendered": stage["R"], "fwd_Msplats_per_s": round(P / (stage["fwd_ms"] * 1e-3) / 1e6, 2),
                "fwd_ms": round(stage["fwd_ms"], 4), "bwd_ms": round(stage["bwd_ms"], 4), "parallelism": "dp%d (views)" % world},
            "roofline": stage["roofline"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, W, H, focal)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def stage_times(B, sd, dL, dev, reps):
    """Forward / backward wall time and the preprocess kernel's duration measured with HIP events on the
    current stream (the stream the library launches on)."""
    empty = torch.Tensor([])
    a = (sd["bg"], sd["means3D"], empty, sd["opacities"], sd["scales"], sd["rotations"], sd["scale_modifier"], empty, empty,
         sd["viewmatrix"], sd["projmatrix"], sd["tanfovx"], sd["tanfovy"], sd["kernel_size"], sd["subpixel_offset"], sd["H"], sd["W"],
         sd["shs"], sd["sh_degree"], sd["campos"], False, False)
    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    fwd, bwd = [], []
    R = 0
    for _ in range(reps):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        R, color, radii, geom, binning, img = B.rasterize_gaussians(*a)
        e1.record()
        B.rasterize_gaussians_backward(a[0], a[1], radii, a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                                       dL, a[17], a[18], a[19], geom, R, binning, img, False)
        e2.record()
        torch.cuda.synchronize()
        fwd.append(e0.elapsed_time(e1)); bwd.append(e1.elapsed_time(e2))
    fwd_ms, bwd_ms = float(np.median(fwd)), float(np.median(bwd))
    # dominant HBM-bound kernel: K1 preprocess (355 B / Gaussian algorithmic: 236 read + 119 written).
    # Timed alone through gof_forward_prepare (K1 + scan + 4-byte read-back) minus nothing: an upper bound on K1's time.
    v = B._View(*a)
    geom = v.bytes_tensor(B.lib.gof_geom_bytes(v.P)); img = v.bytes_tensor(B.lib.gof_image_bytes(v.W, v.H))
    radii = torch.zeros(v.P, dtype=torch.int32, device=dev)
    import ctypes as C
    n = C.c_uint32(0)
    ts = []
    for _ in range(reps):
        e0, e1 = ev(), ev()
        e0.record()
        B._check(B.lib.gof_forward_prepare(v.ref(), B._ptr(geom), geom.numel(), B._ptr(img), img.numel(), B._ptr(radii), C.byref(n), B._stream()))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    k1_ms = float(np.median(ts))
    bytes_k1 = v.P * (236 + 119)
    achieved = bytes_k1 / (k1_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": "preprocess_fwd (+scan, upper bound on its duration)", "achieved": round(achieved, 1), "peak": 8000.0,
            "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": None}
    return {"fwd_ms": fwd_ms, "bwd_ms": bwd_ms, "R": int(R), "roofline": roof}


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
            "sample": "oracle fwd+bwd on %d Gaussians (%.0f%% of the workload's count) at %dx%d, one iteration; fwd %.2fs bwd %.2fs"
                      % (Pc, 100.0 * Pc / args.gaussians, W, H, t1 - t0, t2 - t1)}


if __name__ == "__main__":
    main()
