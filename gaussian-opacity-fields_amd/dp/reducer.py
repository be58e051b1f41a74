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
    def __init__(self, params: Iterable[torch.Tensor], average: bool = False, group=None, wire_dtype=None):
        """wire_dtype: None (default) = all-reduce the fp32 gradients as they are.  torch.bfloat16 halves the bytes on the xGMI
        links (the collective is exposed at the end of the step) at the price of bf16-rounded gradient sums --
        an opt-in for training runs, never used by bench.py's headline."""
        self.params = list(params)
        self.average = average
        self.group = group
        self.wire_dtype = wire_dtype
        self._flat = None
        self._wire = None

    def _ensure(self, n, device, dtype):
        if self._flat is None or self._flat.numel() != n or self._flat.device != device:
            self._flat = torch.empty(n, device=device, dtype=dtype)
        return self._flat

    @staticmethod
    def _shared_bucket(grads):
        """If the gradients are contiguous, ascending, non-overlapping views of ONE storage with < 16 B of padding between them
        (diff_gaussian_rasterization's backward allocates them that way), return the covering 1-D view, else None."""
        g0 = grads[0]
        if any((not g.is_contiguous()) or g.dtype != g0.dtype or g.device != g0.device or
               g.untyped_storage().data_ptr() != g0.untyped_storage().data_ptr() for g in grads):
            return None
        end = g0.storage_offset()
        for g in grads:
            gap = g.storage_offset() - end
            if gap < 0 or gap > 3:
                return None
            end = g.storage_offset() + g.numel()
        start = g0.storage_offset()
        return torch.as_strided(g0, (end - start,), (1,), start)      # padding floats are reduced too (harmless, uninitialised)

    @torch.no_grad()
    def all_reduce(self):
        ps = [p for p in self.params if p.grad is not None]
        if not ps or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        bucket = self._shared_bucket([p.grad for p in ps])
        if bucket is not None and self.wire_dtype is not None:
            if self._wire is None or self._wire.numel() != bucket.numel() or self._wire.device != bucket.device:
                self._wire = torch.empty(bucket.numel(), dtype=self.wire_dtype, device=bucket.device)
            self._wire.copy_(bucket)
            dist.all_reduce(self._wire, op=dist.ReduceOp.SUM, group=self.group)
            bucket.copy_(self._wire)
            if self.average:
                bucket.div_(dist.get_world_size(self.group))
            return
        if bucket is not None:                       # the rasterizer's backward carved them from one allocation: reduce in place
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                bucket.div_(dist.get_world_size(self.group))
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
