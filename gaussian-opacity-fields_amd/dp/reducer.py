"""Gradient all-reduce for view-sharded Gaussian training.

Design for xGMI (point-to-point, 7 links x ~153 GB/s per GPU): ring collectives are per-link
bound, so the 236 B/Gaussian payload is sent as ONE flat fp32 bucket per step (236 MB at 1M
Gaussians: large enough to run at link bandwidth, a single launch instead of six), packed with
one multi-tensor copy.  `_features_rest` is 76 % of the bytes; splitting it into more buckets
only adds launches.  The reduction is a SUM by default (the reference accumulates one view per
optimiser step; summing N views keeps its per-view gradient scale) or a MEAN.

Compressed SH gradient (`sh_params`): the SH gradient of a view is an outer product of 12 B of information per
Gaussian (the blend's colour gradient) with the SH basis of the view direction; ranks all-gather those 12 B and expand
the sum over views locally (gof_sh_grad_pack / gof_sh_grad_expand) instead of all-reducing 192 B.  Bytes on the wire
per Gaussian and rank at N ranks: dense ring all-reduce 2(N-1)/N x 236; compressed 2(N-1)/N x 44 + (N-1) x 12
(N = 8: 413 -> 161; N = 2: 236 -> 56).  The sum runs over the views in rank order: deterministic and identical on all ranks.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


def shard_views(views: List, rank: int, world: int) -> List:
    """Rank r trains on views r, r+world, ... (cams[rank::world])."""
    return list(views[rank::world])


class ViewShards:
    """Which training cameras this rank renders, and the way back: train.py keeps ONE list (train.py:106) for both the per-step
    camera sampling (sharded) and compute_3D_filter (which must see ALL cameras, SURVEY.md 8(e), or the replicas diverge)."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world = rank, world
        self._by_key = {}                    # key (e.g. resolution scale) -> (this rank's cameras, all cameras)

    def shard(self, cams: List, key=1.0) -> List:
        if self.world == 1:
            return cams
        if key not in self._by_key or self._by_key[key][1] is not cams:
            self._by_key[key] = (shard_views(cams, self.rank, self.world), cams)
        return self._by_key[key][0]

    def full(self, cameras):
        """The full list if `cameras` is (a copy of) one of this rank's shards, else `cameras` unchanged."""
        cams = list(cameras)
        for mine, everyone in self._by_key.values():
            if len(cams) == len(mine) and all(a is b for a, b in zip(cams, mine)):
                return everyone
        return cameras


class GradientAllReducer:
    def __init__(self, params: Iterable[torch.Tensor], average: bool = False, group=None, wire_dtype=None, sh_params=None, sh_ops=None,
                 track: bool = True, early_gather: bool = True):
        """sh_params: the SH parameter(s) among `params` whose gradient may travel in compressed form -- one [P,M,3] tensor, or
        the reference's pair (_features_dc [P,1,3], _features_rest [P,M-1,3]).  Used only when their gradient since the last
        exchange stems from exactly one rasterizer backward (else the dense path runs).  sh_ops: the pack / expand
        implementation (default: the HIP kernels behind diff_gaussian_rasterization._backend).  track=False: the caller has
        switched the rasterizer's tracking on itself (a reducer built per step must not reset it).  early_gather: start the
        all-gather of the colour gradient from INSIDE the rasterizer's backward, between its blend and preprocess stages, so that
        the collective runs while preprocess_bwd (and whatever autograd still has to do) executes.
        wire_dtype: None (default) = all-reduce the fp32 gradients as they are.  torch.bfloat16 halves the bytes on the xGMI
        links (the collective is exposed at the end of the step) at the price of bf16-rounded gradient sums --
        an opt-in for training runs, never used by bench.py's headline."""
        self.params = list(params)
        self.average = average
        self.group = group
        self.wire_dtype = wire_dtype
        self._flat = None
        self._wire = None
        self.sh_params = list(sh_params) if sh_params else []
        if len(self.sh_params) > 2:
            raise ValueError("sh_params: one [P,M,3] tensor or the pair (features_dc, features_rest)")
        self._sh_ops = sh_ops
        self.last_exchange = None                    # "dense" | "compressed-sh": what the last all_reduce() did (tests, logging)
        self.last_buckets = None                     # (in-place buckets, packed tensors) of the last dense part (tests, logging)
        self._early = []                             # all-gathers started from inside the backward since the last exchange
        self.profile = False                         # True: stream events around every phase of the exchange (bench.py --gpus N)
        self._phases = []                            # per exchange: {phase: (start event, end event)}
        self._cur = None
        if self.sh_params and track:
            self.enable_sh_tracking(early_gather)

    def enable_sh_tracking(self, early_gather=True):
        ops = self._ops()
        ops.track(True)
        if early_gather and hasattr(ops, "set_ready"):
            ops.set_ready(self._on_colour_gradient)

    def _on_colour_gradient(self, src):
        """Called by the rasterizer's backward after its blend stage.  EVERY tracked backward starts its all-gather here (also a
        second one in the same step, whose result is then discarded): the sequence of collectives stays identical on all ranks."""
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self._early.append(self._start_gather(src, dist.get_world_size(self.group)))

    def _ops(self):
        if self._sh_ops is None:
            from diff_gaussian_rasterization import _backend as B     # HIP library; raises ImportError if it is missing

            class _HipOps:
                track = staticmethod(B.track_sh_grad_source)
                take = staticmethod(B.take_sh_grad_source)
                pack = staticmethod(B.sh_grad_pack)
                expand = staticmethod(B.sh_grad_expand)
                set_ready = staticmethod(B.set_sh_grad_ready_callback)
            self._sh_ops = _HipOps
        return self._sh_ops

    # ---- phase timing (profile=True): events on the current stream; a phase that ends with work.wait() includes the time the
    # compute stream sat waiting for the communication stream, i.e. the EXPOSED part of that collective
    class _Phase:
        def __init__(self, owner, name):
            self.o, self.name = owner, name

        def __enter__(self):
            o = self.o
            if o.profile and torch.cuda.is_available():
                if o._cur is None:
                    o._cur = {}
                self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *exc):
            o = self.o
            if o.profile and torch.cuda.is_available():
                self.e1.record()
                o._cur.setdefault(self.name, []).append((self.e0, self.e1))
            return False

    def _phase(self, name):
        return GradientAllReducer._Phase(self, name)

    def breakdown(self):
        """Mean milliseconds per exchange of every phase recorded since the last call (profile=True): pack (colour-gradient pack +
        camera row), gather_wait (exposed all-gather), expand (SH expansion kernel), bucket_pack / bucket_unpack (copies of gradients
        that do not share the rasterizer's allocation), reduce_wait (exposed all-reduce)."""
        torch.cuda.synchronize()
        out, n = {}, max(1, len(self._phases))
        for ph in self._phases:
            for name, evs in ph.items():
                out[name] = out.get(name, 0.0) + sum(a.elapsed_time(b) for a, b in evs)
        self._phases = []
        return {k: round(v / n, 4) for k, v in out.items()}

    def _start_gather(self, src, world):
        """Pack this view's colour gradient (+ camera centre) and START the all-gather; returns (src, gathered, work)."""
        P = src["P"]
        dev = src["dL_dcolors"].device if "dL_dcolors" in src else src["campos"].device
        mine = torch.empty((P + 1, 3), dtype=torch.float32, device=dev)
        with self._phase("pack"):
            self._ops().pack(src, mine)
            mine[P].copy_(src["campos"].reshape(3))
        gathered = torch.empty((world, P + 1, 3), dtype=torch.float32, device=dev)
        if dist.get_backend(self.group) == "nccl":
            work = dist.all_gather_into_tensor(gathered, mine, group=self.group, async_op=True)
        else:                                        # gloo (CPU tests, single-GPU development runs)
            work = dist.all_gather(list(gathered.unbind(0)), mine, group=self.group, async_op=True)
        return src, gathered, work

    def _sh_begin(self, world):
        """The all-gather of this step's colour gradient -- already in flight if the backward started it, started now otherwise.
        Returns (src, gathered, work, grads), or None when the compressed form is not applicable this step."""
        src = self._ops().take()
        early, self._early = self._early, []
        grads = [p.grad for p in self.sh_params]
        ok = src is not None and not any(g is None for g in grads)
        if ok:
            ok = (sum(g.numel() for g in grads) == 3 * src["M"] * src["P"]
                  and all(g.is_contiguous() and g.dtype == torch.float32 for g in grads))
        if early:
            if ok and len(early) == 1 and early[0][0] is src:
                return early[0] + (grads,)
            for _, _, work in early:                 # not usable (several backwards, foreign gradients): complete them, go dense
                work.wait()
            return None
        if not ok:
            return None
        return self._start_gather(src, world) + (grads,)

    def _sh_finish(self, pending, world):
        """Wait for the all-gather only and expand the sum over views into the SH gradients (the all-reduce of the other
        gradients, issued behind the all-gather, keeps running on the communication stream meanwhile)."""
        src, gathered, work, grads = pending
        with self._phase("gather_wait"):
            work.wait()
        with self._phase("expand"):
            self._ops().expand(src, gathered, 1.0 / world if self.average else 1.0, grads)

    def _ensure(self, n, device, dtype):
        if self._flat is None or self._flat.numel() != n or self._flat.device != device:
            self._flat = torch.empty(n, device=device, dtype=dtype)
        return self._flat

    @staticmethod
    def _shared_bucket(grads):
        """If the gradients are contiguous, ascending, non-overlapping views of ONE storage with < 16 B of padding between them
        (diff_gaussian_rasterization's backward allocates them that way), return the covering 1-D view, else None."""
        grads = sorted(grads, key=lambda g: g.storage_offset())
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
            if self.sh_params:
                self._ops().take()                   # nothing to exchange: forget the tracked backward
            return
        world = dist.get_world_size(self.group)
        self.last_exchange = "dense"
        pending = self._sh_begin(world) if self.sh_params else None
        if pending is not None:
            self.last_exchange = "compressed-sh"
            sh_ids = {id(p) for p in self.sh_params}
            ps = [p for p in ps if id(p) not in sh_ids]
        finish_dense = self._dense_begin(ps, world) if ps else None
        if pending is not None:
            self._sh_finish(pending, world)
        if finish_dense is not None:
            finish_dense()
        if self.profile and self._cur is not None:
            self._phases.append(self._cur)
            self._cur = None

    def _dense_begin(self, ps, world):
        """Start the all-reduce of the gradients of `ps`; returns the callable that completes it.  Gradients that are views of ONE
        allocation without gaps -- the rasterizer's backward carves the parameter gradients that way, and under
        train_epilogue.activations.INPLACE_GRAD the activation backwards keep them there -- are reduced IN PLACE, one collective per
        such allocation; whatever is left (appearance network, foreign gradients) is packed into one flat buffer."""
        by_storage = {}
        for p in ps:
            g = p.grad
            key = (g.untyped_storage().data_ptr(), g.device, g.dtype) if g.is_contiguous() else None
            by_storage.setdefault(key, []).append(p)
        buckets, rest = [], []
        for key, group in by_storage.items():
            b = self._shared_bucket([p.grad for p in group]) if (key is not None and self.wire_dtype is None) else None
            if b is not None and (len(group) > 1 or len(by_storage) == 1):
                buckets.append(b)
            else:
                rest.extend(group)
        if self.wire_dtype is not None and len(by_storage) == 1 and None not in by_storage:
            bucket = self._shared_bucket([p.grad for p in ps])
            if bucket is not None:
                if self._wire is None or self._wire.numel() != bucket.numel() or self._wire.device != bucket.device:
                    self._wire = torch.empty(bucket.numel(), dtype=self.wire_dtype, device=bucket.device)
                self._wire.copy_(bucket)
                work = dist.all_reduce(self._wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.last_buckets = (1, 0)

                def finish_wire():
                    with self._phase("reduce_wait"):
                        work.wait()
                    bucket.copy_(self._wire)
                    if self.average:
                        bucket.div_(world)
                return finish_wire
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in buckets]
        flat = views = grads = None
        if rest:
            for p in rest:                           # a non-contiguous gradient would be COPIED by reshape(-1) and the reduced values lost
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
            grads = [p.grad.view(-1) for p in rest]
            n = sum(g.numel() for g in grads)
            flat = self._ensure(n, grads[0].device, grads[0].dtype)
            views = list(torch.split(flat, [g.numel() for g in grads]))
            with self._phase("bucket_pack"):
                torch._foreach_copy_(views, grads)
            wire = flat
            if self.wire_dtype is not None:          # opt-in reduced-precision wire format, also when the gradients had to be packed
                if self._wire is None or self._wire.numel() != n or self._wire.device != flat.device:
                    self._wire = torch.empty(n, dtype=self.wire_dtype, device=flat.device)
                wire = self._wire
                wire.copy_(flat)
            works.append(dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.last_buckets = (len(buckets), len(rest))

        def finish():
            with self._phase("reduce_wait"):
                for w in works:
                    w.wait()
            if self.average:
                for b in buckets:
                    b.div_(world)
            if rest:
                if wire is not flat:
                    flat.copy_(wire)
                if self.average:
                    flat.div_(world)
                with self._phase("bucket_unpack"):
                    torch._foreach_copy_(grads, views)
        return finish


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
