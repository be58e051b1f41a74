"""GPU, two ranks on ONE device (gloo carries the collectives; the driver's multi-GPU runs use RCCL with the same code path):
the data-parallel training step with the real kernels -- rasterizer forward/backward per rank on its own view, then the gradient
exchange of dp.GradientAllReducer in both forms (dense all-reduce of everything; SH gradient compressed to 12 B per Gaussian and
view + all-reduce of the rest).  Both must leave identical gradients on both ranks."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synthetic_scenes as S
    from gpu_common import to_dev, settings_from
    from diff_gaussian_rasterization import GaussianRasterizer
    from dp import GradientAllReducer
    from test_parity_gpu import _orbit_camera
    base = S.scene_lego_like(P=20000, W=200, H=150, seed=8)                   # identical replica on every rank
    base["opacities"][:] = 0.5
    cam = _orbit_camera(200, 150, 0.69, 0.7 + 2.1 * rank, 0.5 - 0.6 * rank)    # one view per rank
    sd = to_dev({**base, **cam})
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    from diff_gaussian_rasterization import _backend as B
    # ONE forward/backward per rank (the backward's atomic accumulation order differs from run to run, so every exchange form is
    # applied to copies of the same local gradients).  The first reducer is the production configuration: tracking on and the
    # all-gather started from INSIDE the backward, between its blend and preprocess stages.
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    red_early = GradientAllReducer(list(params.values()), sh_params=[params["shs"]])
    color, _ = GaussianRasterizer(settings_from(sd))(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                                     opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    dL = torch.randn(color.shape, generator=torch.Generator().manual_seed(3 + rank)).to(color.device)
    color.backward(dL)
    started_early = len(red_early._early) == 1
    local = {k: params[k].grad.clone() for k in names}
    src = B._sh_track["src"]
    red_early.all_reduce()
    torch.cuda.synchronize()
    results = {"early": ({k: params[k].grad.clone() for k in names}, red_early.last_exchange, local["shs"])}
    B.set_sh_grad_ready_callback(None)
    for mode in ("dense", "compressed", "compressed-mean"):
        for k in names:
            params[k].grad = local[k].clone()
        red = GradientAllReducer(list(params.values()), sh_params=None if mode == "dense" else [params["shs"]], average=mode.endswith("mean"),
                                 track=False)
        B._sh_track.update(count=1, src=src)                                   # as if the backward had just run
        red.all_reduce()
        torch.cuda.synchronize()
        results[mode] = ({k: params[k].grad.clone() for k in names}, red.last_exchange, local["shs"])
    B.track_sh_grad_source(False)
    ok = results["dense"][1] == "dense" and results["compressed"][1] == "compressed-sh" and results["early"][1] == "compressed-sh" and started_early
    msg = []
    for k in names:
        if not torch.equal(results["early"][0][k], results["dense"][0][k]):
            ok = False; msg.append("%s: gather started inside the backward vs dense" % k)
    for k in names:
        a, b, c = results["dense"][0][k], results["compressed"][0][k], results["compressed-mean"][0][k]
        # two ranks: a + b is the only possible order, so the two forms agree bit for bit
        if not torch.equal(a, b):
            ok = False; msg.append("%s: dense vs compressed max diff %.3e" % (k, (a - b).abs().max().item()))
        if not torch.allclose(c, a / world, rtol=1e-6, atol=0):
            ok = False; msg.append("%s: mean" % k)
    # the exchange really added the other rank's view: the reduced SH gradient differs from the local one
    ok = ok and not torch.equal(results["compressed"][0]["shs"], results["compressed"][2])
    # every rank holds the same reduced gradients
    chk = torch.stack([results["compressed"][0][k].double().sum() for k in names]).cpu()
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    ok = ok and all(torch.equal(g, gathered[0]) for g in gathered)
    ret[rank] = (bool(ok), "; ".join(msg))
    dist.destroy_process_group()


def test_two_ranks_dense_and_compressed_gradient_exchange_agree():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(ret.get(r, (False, "no result"))[0] for r in range(world)), dict(ret)


def _worker_rccl(port, ret):
    """One rank, backend "nccl" (= RCCL): world size 1 makes every collective an identity, so the exchange must hand back the local
    gradients bit for bit -- while the calls themselves (all_gather_into_tensor of the packed colour gradient, the in-place
    all-reduce of the shared gradient bucket, the packed fallback) go through RCCL on the GPU exactly as they do at N > 1."""
    for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import synthetic_scenes as S
    from gpu_common import to_dev, settings_from
    from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
    from dp import GradientAllReducer
    sd = to_dev(S.scene_lego_like(P=20000, W=200, H=150, seed=8))
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)
    red = GradientAllReducer(list(params.values()), sh_params=[params["shs"]])
    color, _ = GaussianRasterizer(settings_from(sd))(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                                     opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    color.backward(torch.randn(color.shape, generator=torch.Generator().manual_seed(3)).to(color.device))
    local = {k: params[k].grad.clone() for k in names}
    msg = []
    assert dist.get_backend() == "nccl"
    # compressed form: all-gather of 12 B per Gaussian over RCCL + local expansion, all-reduce of the rest in place
    pending = red._sh_begin(1)
    if pending is None:
        msg.append("the compressed SH exchange was not applicable")
    else:
        rest = [params[k] for k in names if k != "shs"]
        finish = red._dense_begin(rest, 1)
        red._sh_finish(pending, 1)
        finish()
        torch.cuda.synchronize()
        for k in names:
            if not torch.equal(params[k].grad, local[k]):
                msg.append("compressed: %s changed (max diff %.3e)" % (k, (params[k].grad - local[k]).abs().max().item()))
    # dense form over everything: the rasterizer's one allocation reduced in place
    bucket = red._shared_bucket([params[k].grad for k in names])
    if bucket is None:
        msg.append("the gradients of one backward are not one bucket")
    red._dense_begin([params[k] for k in names], 1)()
    torch.cuda.synchronize()
    for k in names:
        if not torch.equal(params[k].grad, local[k]):
            msg.append("dense: %s changed" % k)
    # frames whose backward starts the exchange run on the same optimistic pools as on one GPU (round 5): the blend stage is verified
    # against the forward's counters BEFORE the colour gradient is handed to the reducer, so nothing is ever repeated after it has gone
    # on the wire -- and there is no synchronising read-back in front of the backward on the sync-free forward's frames
    if not B._exchange_starts_inside_backward():
        msg.append("the reducer's ready-callback is not installed")
    q0 = B._stats["backward_queries"]
    for _ in range(3):
        color, _ = GaussianRasterizer(settings_from(sd))(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                                         opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
        color.sum().backward()
    torch.cuda.synchronize()
    if B._stats["backward_queries"] != q0:
        msg.append("%d read-backs in front of backwards that start the exchange" % (B._stats["backward_queries"] - q0))
    key = [k for k in B._capacity if k[1] == sd["means3D"].shape[0]]
    if not key or B._mask_pool_subchunks(key[0]) is None:
        msg.append("the frames of a data-parallel step still run on a worst-case mask pool")
    red._early = []
    B._sh_track.update(count=0, src=None)
    # packed fallback (gradients that are separate allocations), averaged: still the identity at one rank
    for k in names:
        params[k].grad = local[k].clone()
    red_avg = GradientAllReducer(list(params.values()), average=True, track=False)
    red_avg._dense_begin([params[k] for k in names], 1)()
    torch.cuda.synchronize()
    for k in names:
        if not torch.equal(params[k].grad, local[k]):
            msg.append("fallback: %s changed" % k)
    B.set_sh_grad_ready_callback(None)
    B.track_sh_grad_source(False)
    ret[0] = (not msg, "; ".join(msg))
    dist.destroy_process_group()


def test_the_exchange_runs_over_rccl_on_one_rank_and_is_the_identity():
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_worker_rccl, args=(port, ret))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert ret.get(0, (False, "no result"))[0], dict(ret)


def test_inplace_activation_backwards_leave_the_parameter_gradients_in_the_rasterizers_bucket():
    """launch/run_train_dp.py's configuration (train_epilogue.activations.INPLACE_GRAD): raw parameters -> HIP activations through
    the property bodies the launcher installs on GaussianModel (3D-filter opacity + scaling as ONE autograd node, normalised
    rotation, in render()'s order) -> rasterizer -> backward.  The gradients of _xyz,
    _opacity, _scaling, _rotation are then views of ONE allocation (the rasterizer's gradient bucket) that the reducer reduces in
    place -- and have the same values as with fresh gradient tensors (bit for bit: the same kernels, only the output address
    differs; the scaling gradient's two contributions are added in the same order)."""
    import importlib
    for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import synthetic_scenes as S
    from gpu_common import to_dev, settings_from
    from diff_gaussian_rasterization import GaussianRasterizer
    from dp import GradientAllReducer
    act = importlib.import_module("train_epilogue.activations")
    sc = S.scene_frustum(30_000, W=320, H=208, focal=240.0, seed=2, pose_seed=3)
    sd = to_dev(sc)
    g = torch.Generator().manual_seed(4)
    raw0 = {"xyz": sd["means3D"].clone(), "opacity": torch.randn(30_000, 1, generator=g).cuda(), "scaling": torch.log(sd["scales"]),
            "rotation": (sd["rotations"] * 1.7).contiguous()}
    filter_3D = (sd["scales"].min(dim=1, keepdim=True).values * 0.3).contiguous()
    dL = torch.randn((9, sd["H"], sd["W"]), generator=g).cuda()

    def run(inplace):
        act.INPLACE_GRAD = inplace
        try:
            raw = {k: v.clone().requires_grad_(True) for k, v in raw0.items()}
            shs = sd["shs"].clone().requires_grad_(True)
            means2D = torch.zeros_like(raw["xyz"], requires_grad=True)
            class Model:                          # the attributes the property bodies read (scene/gaussian_model.py)
                _opacity, _scaling, _rotation = raw["opacity"], raw["scaling"], raw["rotation"]
            Model.filter_3D = filter_3D
            m = Model()
            opacity = act.get_opacity_with_3D_filter(m)                                         # render(): opacity first ...
            scales = act.get_scaling_with_3D_filter(m)                                          # ... then scales, rotations
            rot = act.get_rotation(m)
            color, _ = GaussianRasterizer(settings_from(sd))(means3D=raw["xyz"], means2D=means2D, shs=shs, opacities=opacity, scales=scales, rotations=rot)
            color.backward(dL)
            torch.cuda.synchronize()
            return raw
        finally:
            act.INPLACE_GRAD = False
    a, b = run(False), run(True)
    grads_b = [b[k].grad for k in ("xyz", "opacity", "scaling", "rotation")]
    bucket = GradientAllReducer._shared_bucket(grads_b)
    assert bucket is not None and all(x.untyped_storage().data_ptr() == bucket.untyped_storage().data_ptr() for x in grads_b)
    assert GradientAllReducer._shared_bucket([a[k].grad for k in ("xyz", "opacity", "scaling", "rotation")]) is None      # fresh tensors: would be packed
    for k in ("xyz", "opacity", "scaling", "rotation"):
        assert torch.isfinite(b[k].grad).all() and b[k].grad.abs().max() > 0
        assert torch.equal(a[k].grad, b[k].grad), k
