"""Gradient all-reduce for view-sharded Gaussian training.

Design for xGMI (point-to-point, 7 links x ~153 GB/s per GPU): ring collectives are per-link
bound, so the 236 B/Gaussian payload is sent as ONE flat fp32 bucket per step (236 MB at 1M
Gaussians: large enough to run at link bandwidth, a single launch instead of six), packed with
one multi-tensor copy.  `_features_rest` is 76 % of the bytes; splitting it into more buckets
only adds launches.  The reduction is a SUM by default (the reference accumulates one view per
optimiser step; summing N views keeps its per-view gradient scale) or a MEAN.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


def shard_views(views: List, rank: int, world: int) -> List:
    """Rank r trains on views r, r+world, ... (cams[rank::world])."""
    return list(views[rank::world])


class GradientAllReducer:
    def __init__(self, params: Iterable[torch.Tensor], average: bool = False, group=None):
        self.params = list(params)
        self.average = average
        self.group = group
        self._flat = None

    def _ensure(self, n, device, dtype):
        if self._flat is None or self._flat.numel() != n or self._flat.device != device:
            self._flat = torch.empty(n, device=device, dtype=dtype)
        return self._flat

    @torch.no_grad()
    def all_reduce(self):
        ps = [p for p in self.params if p.grad is not None]
        if not ps or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        grads = [p.grad.reshape(-1) for p in ps]
        n = sum(g.numel() for g in grads)
        flat = self._ensure(n, grads[0].device, grads[0].dtype)
        views = list(torch.split(flat, [g.numel() for g in grads]))
        torch._foreach_copy_(views, grads)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        if self.average:
            flat.div_(dist.get_world_size(self.group))
        torch._foreach_copy_(grads, views)


def all_reduce_densification_stats(xyz_gradient_accum, xyz_gradient_accum_abs, denom, max_radii2D, xyz_gradient_accum_abs_max=None,
                                   group=None):
    """Make densify_and_prune identical on every rank (scene/gaussian_model.py:709-714, train.py:255-264):
    SUM the accumulators, MAX the radii / abs-max statistics."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in (xyz_gradient_accum, xyz_gradient_accum_abs, denom):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    for t in (max_radii2D, xyz_gradient_accum_abs_max):
        if t is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
