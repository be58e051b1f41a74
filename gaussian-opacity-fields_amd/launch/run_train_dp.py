#!/usr/bin/env python3
"""Data-parallel launcher for the reference's UNCHANGED train.py: one process per GPU, cameras sharded
across ranks, Gaussian parameter gradients all-reduced over RCCL before every optimizer step.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        gaussian-opacity-fields_amd/launch/run_train_dp.py /path/to/gaussian-opacity-fields/train.py -s <scene> ...

Injection points; NEW behaviour, the reference has no multi-GPU training:
  * train.py:370 pins cuda:0           -> HIP_VISIBLE_DEVICES=<LOCAL_RANK> is exported before torch initialises HIP
  * train.py:135-137 camera sampling   -> Scene.getTrainCameras returns this rank's shard cams[rank::world]
  * gaussian_model.py:342-364          -> training_setup registers an optimizer step pre-hook that all-reduces
                                          the parameter gradients (the optimizer object survives densification); the SH
                                          gradient (81 % of the bytes) travels in compressed form, dp/reducer.py
                                          (GOF_DP_DENSE_SH=1: dense all-reduce of everything)
  * gaussian_model.py:262-311          -> compute_3D_filter is fed ALL training cameras on every rank, not the shard train.py holds
  * gaussian_model.py:685-707          -> densify_and_prune first all-reduces the statistics accumulated since the last
                                          densification (SUM for the accumulators / denom, MAX for max_radii2D / abs-max)
  * train.py:247-250,276-301           -> only rank 0 writes point clouds / checkpoints / TensorBoard
  * train.py:108-110, 157-159          -> --use_decoupled_appearance: train.py numbers the cameras it holds (camera.idx) and indexes the
                                          per-camera appearance embedding with that number; here every camera carries its index in
                                          the FULL (unsharded) train + test list, so all ranks address the same embedding rows and
                                          the all-reduced embedding / network gradients mean the same thing everywhere
Identical seeds on every rank (train.py:367-369) keep densify_and_split's sampling identical.

Semantics of one optimiser step with N ranks: N views are rendered (one per rank), their parameter gradients
are SUMMED (GOF_DP_AVERAGE=1: averaged -- Adam's update is invariant to the scale up to its eps = 1e-15, so the two differ only in
what a logged gradient norm means) and every rank applies the identical Adam step.  `--iterations` keeps counting optimiser steps:
a run of I steps consumes N x I views.  The iteration-indexed schedules of train.py (densification window and interval, opacity
reset, regulariser start, position-LR decay length, test / save / checkpoint iterations) are NOT rescaled by default;
`--gof_views_per_step_schedule` divides every one of them by N on the command line handed to train.py, so that the run sees the same
number of VIEWS per schedule stage as the reference's single-GPU recipe (two things train.py hard-codes stay per step: the SH degree
rises every 1000 steps, train.py:131, and the late 3D-filter refresh runs every 100 steps, train.py:266).  Every rank runs
prepare_output_and_logger (identical cfg_args; TensorBoard, if present, logs rank-local losses) -- only rank 0 saves models.

GOF_DP_SHARE_GPU=1 (development / single-GPU tests): all ranks use the visible GPU(s) round-robin and gloo carries the collectives.
"""
import os
import sys

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
SHARE_GPU = os.environ.get("GOF_DP_SHARE_GPU") == "1"


def pick_visible_device(local_rank, env, share=False):
    """The HIP_VISIBLE_DEVICES value for this rank (train.py:370 pins cuda:0, so every rank must see exactly its GPU as device 0).
    A mask the user or scheduler already set (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES, e.g. "4,5,6,7") is
    indexed, not overwritten."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        vis = [v for v in env.get(var, "").split(",") if v.strip() != ""]
        if vis:
            if share:
                return vis[local_rank % len(vis)]
            if local_rank >= len(vis):
                raise RuntimeError("run_train_dp.py: local rank %d but %s=%s lists only %d device(s)" % (local_rank, var, env[var], len(vis)))
            return vis[local_rank]
    return "0" if share else str(local_rank)           # ROCR_VISIBLE_DEVICES (if set) re-numbers the devices HIP sees: index into that numbering


os.environ["HIP_VISIBLE_DEVICES"] = pick_visible_device(local_rank, os.environ, SHARE_GPU)   # before importing torch
os.environ.pop("CUDA_VISIBLE_DEVICES", None)              # one mask only: the two would be intersected
# RCCL between processes that each see ONE device: peer buffers are exchanged as dmabuf IPC handles, the only form this platform's
# driver supports (without it: hipIpcGetMemHandle "invalid argument" at communicator set-up).  Nothing else of RCCL's environment is
# touched -- channel counts (NCCL_MIN_NCHANNELS) and protocol choices stay at the library's defaults until an 8-GPU run measures them
# (DESIGN.md section 6); whatever the user exports is passed through.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, PKG)


def main():
    import torch
    import torch.distributed as dist
    from dp import GradientAllReducer, ViewShards
    from dp.reducer import all_reduce_densification_stats

    script = os.path.abspath(sys.argv[1])
    sys.path.insert(0, os.path.dirname(script))
    sys.path.insert(0, PKG)
    sys.path.append(os.path.join(PKG, "shims"))
    if SHARE_GPU:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    rank, world = dist.get_rank(), dist.get_world_size()
    dist.all_reduce(torch.zeros(1, device="cuda"))    # the communicator exists before a collective is issued from the autograd thread
    if "--gof_views_per_step_schedule" in sys.argv:
        sys.argv.remove("--gof_views_per_step_schedule")
        sys.argv[2:] = scale_schedule_args(sys.argv[2:], world)

    import scene as ref_scene
    from scene.gaussian_model import GaussianModel

    _get_train = ref_scene.Scene.getTrainCameras

    shards = ViewShards(rank, world)

    def get_train_sharded(self, scale=1.0):
        cams = _get_train(self, scale)
        if len(cams) < world:
            # an empty shard would make train.py's randint(0, -1) raise on that rank only, with the others blocked in collectives
            raise RuntimeError("run_train_dp.py: %d training camera(s) for %d ranks -- every rank needs at least one view" % (len(cams), world))
        # camera.idx (train.py:108-110) = position in the FULL train + test list, whatever list train.py enumerates on this rank
        for i, cam in enumerate(list(cams) + list(self.getTestCameras(scale))):
            cam.__dict__["_gof_global_idx"] = i
        return shards.shard(cams, scale)
    ref_scene.Scene.getTrainCameras = get_train_sharded
    from scene.cameras import Camera
    Camera.idx = property(lambda self: self.__dict__["_gof_global_idx"], lambda self, value: None)    # train.py's per-shard numbering is ignored

    # compute_3D_filter(cameras=trainCameras) (train.py:118,261,269) receives the shard; the filter depends on ALL training cameras
    # (per-point minimum depth over the cameras that see it, gaussian_model.py:262-311) and must be identical on every rank: map
    # the shard back to the full list -- every rank holds all camera poses, no communication needed.
    full_camera_list = shards.full
    import importlib
    # (not `import train_epilogue.filter_3d as m`: the package re-exports a FUNCTION of that name, which such an import would bind)
    _f3d = importlib.import_module("train_epilogue.filter_3d")
    assert hasattr(_f3d, "compute_3D_filter") and hasattr(_f3d, "CAMERA_LIST_HOOK")
    _f3d.CAMERA_LIST_HOOK = full_camera_list
    _ref_filter = GaussianModel.compute_3D_filter          # effective with GOF_TORCH_EPILOGUE=1 (the reference's own method stays)
    GaussianModel.compute_3D_filter = lambda self, cameras: _ref_filter(self, full_camera_list(cameras))

    _setup = GaussianModel.training_setup
    # the activation backwards leave the gradients of _opacity / _scaling / _rotation in the rasterizer's gradient allocation, next
    # to _xyz's: the reducer then all-reduces that bucket in place (train_epilogue/activations.py; GOF_DP_INPLACE_ACT=0 switches it off)
    if os.environ.get("GOF_DP_INPLACE_ACT", "1") != "0" and os.environ.get("GOF_TORCH_EPILOGUE") != "1":
        import importlib
        importlib.import_module("train_epilogue.activations").INPLACE_GRAD = True

    def training_setup(self, training_args):
        _setup(self, training_args)

        dense = os.environ.get("GOF_DP_DENSE_SH") == "1"
        reducer = GradientAllReducer([], track=False, average=os.environ.get("GOF_DP_AVERAGE") == "1")
        check_every = int(os.environ.get("GOF_DP_CHECK_EVERY", "0"))      # replica-consistency check (debugging / tests): 0 = off
        steps = [0]

        def pre_step(optimizer, args, kwargs):
            # the nn.Parameters are replaced by every densification (gaussian_model.py:532-607): collect them per step
            reducer.params = [p for g in optimizer.param_groups for p in g["params"]]
            by_name = {g.get("name"): g["params"][0] for g in optimizer.param_groups if len(g["params"]) == 1}
            sh = [by_name[n] for n in ("f_dc", "f_rest") if n in by_name]      # gaussian_model.py:351-352
            reducer.sh_params = sh if (len(sh) == 2 and not dense) else []
            reducer.all_reduce()
            steps[0] += 1
            if check_every and steps[0] % check_every == 0:
                check_replicas(optimizer, steps[0], {"filter_3D": getattr(self, "filter_3D", None)})
        self.optimizer.register_step_pre_hook(pre_step)
        if not dense:
            reducer.enable_sh_tracking()          # + the all-gather starts inside the rasterizer's backward
    GaussianModel.training_setup = training_setup

    # Densification statistics are accumulated per rank (each rank sees other views) and zeroed by every densification
    # (densification_postfix, gaussian_model.py:609-629).  Reducing the running totals ONCE, right before they are consumed,
    # gives every rank the statistics of all views since the last densification -> identical densify / prune decisions.
    def densify_and_prune(self, *a, **k):
        all_reduce_densification_stats(self.xyz_gradient_accum, self.xyz_gradient_accum_abs, self.denom,
                                       self.max_radii2D, getattr(self, "xyz_gradient_accum_abs_max", None))
        return densify_and_prune.inner(self, *a, **k)
    # `inner` is what actually densifies: the reference's method now, the device implementation once run_reference_script.py has
    # rebound the training epilogue (it looks for this attribute instead of overwriting the wrapper)
    densify_and_prune.inner = GaussianModel.densify_and_prune
    GaussianModel.densify_and_prune = densify_and_prune

    if rank != 0:      # only rank 0 writes
        ref_scene.Scene.save = lambda self, iteration: None
        _save = torch.save
        torch.save = lambda *a, **k: None

    import runpy
    sys.argv = [os.path.join(PKG, "launch", "run_reference_script.py"), script] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")


def check_replicas(optimizer, step, extra=None):
    """Every rank must hold bit-identical parameters and (after the exchange) bit-identical gradients: compare order-independent
    integer checksums of their bit patterns across ranks and name the first tensor that differs."""
    import torch
    import torch.distributed as dist
    names, sums = [], []
    for g in optimizer.param_groups:
        for i, p in enumerate(g["params"]):
            for kind, t in (("param", p), ("grad", p.grad)):
                if t is None:
                    continue
                names.append("%s[%d].%s%s" % (g.get("name", "?"), i, kind, tuple(t.shape)))
                sums.append(t.detach().contiguous().view(torch.int32).to(torch.int64).sum())
    for name, t in (extra or {}).items():            # replicated state outside the optimizer (the 3D filter)
        if t is not None:
            names.append("%s%s" % (name, tuple(t.shape)))
            sums.append(t.detach().contiguous().view(torch.int32).to(torch.int64).sum())
    mine = torch.stack(sums + [torch.tensor(len(sums), device=sums[0].device)])
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        everyone = torch.empty((world,) + mine.shape, dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(everyone, mine)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        everyone = torch.stack(parts)
    everyone = everyone.cpu()
    for r in range(1, world):
        if not torch.equal(everyone[r], everyone[0]):
            bad = [n for n, a, b in zip(names, everyone[0].tolist(), everyone[r].tolist()) if a != b]
            raise RuntimeError("run_train_dp.py: replicas diverged at optimiser step %d: rank %d differs from rank 0 in %s" % (step, r, bad[:6] or "the tensor count"))


SCHEDULE_OPTIONS = ("--iterations", "--position_lr_max_steps", "--densify_from_iter", "--densify_until_iter", "--densification_interval",
                    "--opacity_reset_interval", "--distortion_from_iter", "--depth_normal_from_iter")
SCHEDULE_LISTS = ("--test_iterations", "--save_iterations", "--checkpoint_iterations")
SCHEDULE_DEFAULTS = {"--iterations": 30_000, "--position_lr_max_steps": 30_000, "--densify_from_iter": 500, "--densify_until_iter": 15_000,
                     "--densification_interval": 100, "--opacity_reset_interval": 3000, "--distortion_from_iter": 15_000,
                     "--depth_normal_from_iter": 15_000}      # arguments/__init__.py:79-101


def scale_schedule_args(argv, world):
    """Divide every iteration-indexed option of train.py by the number of views per step (options that are absent are added with
    the reference's default divided the same way), so that a schedule stage lasts the same number of VIEWS as in the reference."""
    argv = list(argv)
    div = lambda v: max(1, int(round(int(v) / world)))     # noqa: E731
    seen = set()
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in SCHEDULE_OPTIONS and i + 1 < len(argv):
            argv[i + 1] = str(div(argv[i + 1]))
            seen.add(a)
            i += 2
        elif a in SCHEDULE_LISTS:
            i += 1
            while i < len(argv) and not argv[i].startswith("-"):
                argv[i] = str(div(argv[i]))
                i += 1
        else:
            i += 1
    for a, v in SCHEDULE_DEFAULTS.items():
        if a not in seen:
            argv += [a, str(div(v))]
    if not any(a in argv for a in ("--test_iterations",)):
        argv += ["--test_iterations", str(div(7000)), str(div(30000))]
    if "--save_iterations" not in argv:
        argv += ["--save_iterations", str(div(7000)), str(div(30000))]
    return argv


if __name__ == "__main__":
    main()
