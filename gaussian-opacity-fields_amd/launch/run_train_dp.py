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
Identical seeds on every rank (train.py:367-369) keep densify_and_split's sampling identical.
"""
import os
import sys

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
os.environ["HIP_VISIBLE_DEVICES"] = str(local_rank)       # before importing torch: every rank sees its GPU as cuda:0

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, PKG)


def main():
    import torch
    import torch.distributed as dist
    from dp import GradientAllReducer, ViewShards
    from dp.reducer import all_reduce_densification_stats

    script = os.path.abspath(sys.argv[1])
    if "--use_decoupled_appearance" in sys.argv:
        # train.py:109-110 numbers the cameras it holds (camera.idx) and indexes the appearance embeddings with that number; under
        # view sharding every rank would number its own shard, so the all-reduced embedding gradients would mix different cameras
        raise NotImplementedError("run_train_dp.py: --use_decoupled_appearance is not supported with view sharding (per-camera "
                                  "embedding indices are assigned per rank by train.py:109-110)")
    sys.path.insert(0, os.path.dirname(script))
    sys.path.insert(0, PKG)
    sys.path.append(os.path.join(PKG, "shims"))
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    rank, world = dist.get_rank(), dist.get_world_size()
    dist.all_reduce(torch.zeros(1, device="cuda"))    # the communicator exists before a collective is issued from the autograd thread

    import scene as ref_scene
    from scene.gaussian_model import GaussianModel

    _get_train = ref_scene.Scene.getTrainCameras

    shards = ViewShards(rank, world)

    def get_train_sharded(self, scale=1.0):
        return shards.shard(_get_train(self, scale), scale)
    ref_scene.Scene.getTrainCameras = get_train_sharded

    # compute_3D_filter(cameras=trainCameras) (train.py:118,261,269) receives the shard; the filter depends on ALL training cameras
    # (per-point minimum depth over the cameras that see it, gaussian_model.py:262-311) and must be identical on every rank: map
    # the shard back to the full list -- every rank holds all camera poses, no communication needed.
    full_camera_list = shards.full
    import train_epilogue.filter_3d as _f3d
    _f3d.CAMERA_LIST_HOOK = full_camera_list
    _ref_filter = GaussianModel.compute_3D_filter          # effective with GOF_TORCH_EPILOGUE=1 (the reference's own method stays)
    GaussianModel.compute_3D_filter = lambda self, cameras: _ref_filter(self, full_camera_list(cameras))

    _setup = GaussianModel.training_setup

    def training_setup(self, training_args):
        _setup(self, training_args)

        dense = os.environ.get("GOF_DP_DENSE_SH") == "1"
        reducer = GradientAllReducer([], track=False)

        def pre_step(optimizer, args, kwargs):
            # the nn.Parameters are replaced by every densification (gaussian_model.py:532-607): collect them per step
            reducer.params = [p for g in optimizer.param_groups for p in g["params"]]
            by_name = {g.get("name"): g["params"][0] for g in optimizer.param_groups if len(g["params"]) == 1}
            sh = [by_name[n] for n in ("f_dc", "f_rest") if n in by_name]      # gaussian_model.py:351-352
            reducer.sh_params = sh if (len(sh) == 2 and not dense) else []
            reducer.all_reduce()
        self.optimizer.register_step_pre_hook(pre_step)
        if not dense:
            reducer.enable_sh_tracking()          # + the all-gather starts inside the rasterizer's backward
    GaussianModel.training_setup = training_setup

    # Densification statistics are accumulated per rank (each rank sees other views) and zeroed by every densification
    # (densification_postfix, gaussian_model.py:609-629).  Reducing the running totals ONCE, right before they are consumed,
    # gives every rank the statistics of all views since the last densification -> identical densify / prune decisions.
    _densify = GaussianModel.densify_and_prune

    def densify_and_prune(self, *a, **k):
        all_reduce_densification_stats(self.xyz_gradient_accum, self.xyz_gradient_accum_abs, self.denom,
                                       self.max_radii2D, getattr(self, "xyz_gradient_accum_abs_max", None))
        return _densify(self, *a, **k)
    GaussianModel.densify_and_prune = densify_and_prune

    if rank != 0:      # only rank 0 writes
        ref_scene.Scene.save = lambda self, iteration: None
        _save = torch.save
        torch.save = lambda *a, **k: None

    import runpy
    sys.argv = [os.path.join(PKG, "launch", "run_reference_script.py"), script] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
