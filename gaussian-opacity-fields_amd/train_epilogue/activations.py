"""Fused replacements for the three derived tensors ``render()`` reads from the reference's ``GaussianModel`` every iteration
(gaussian_renderer/__init__.py:60,70-71):

    get_scaling_with_3D_filter  (scene/gaussian_model.py:157-162)   5 torch kernels + autograd  -> 1 + 1 launches
    get_opacity_with_3D_filter  (:183-194)                         10 torch kernels + autograd  -> 1 + 1 launches
    get_rotation                (:165-166, F.normalize)             3 torch kernels + autograd  -> 1 + 1 launches

launch/run_reference_script.py installs them as properties on the reference's GaussianModel class.  Same values to fp32
rounding, same gradients (tests/test_train_epilogue_gpu.py)."""
import ctypes as C

import torch

from . import _backend as B

lib = B.lib
_vp, _i64 = C.c_void_p, C.c_int64

# Data-parallel training (launch/run_train_dp.py sets this): the backward of an activation writes the gradient of the RAW parameter
# over its incoming gradient -- same shape, element-wise kernels -- instead of into a fresh tensor.  The incoming gradients are the
# rasterizer's dL_dopacity / dL_dscales / dL_drotations, segments of the ONE allocation its backward carves the parameter gradients
# from (diff_gaussian_rasterization/_backend.py), so _opacity.grad / _scaling.grad / _rotation.grad end up next to _xyz.grad in that
# allocation and dp.GradientAllReducer all-reduces them IN PLACE (no pack / unpack of 44 B per Gaussian).  _scaling receives a second
# contribution from the opacity activation, and autograd would add two contributions out of place (a view of a shared allocation is
# never accumulated into in place: torch/csrc/autograd/input_buffer.cpp); so under this flag the two activations that read _scaling
# are ONE autograd node (_ScalingOpacity: the property evaluated first computes both, the other takes its result), whose backward
# adds the opacity path's contribution to the scaling gradient itself, in place.  If a gradient still ends up elsewhere the reducer
# simply packs that tensor: correctness does not depend on any of this.  Off by default, and supported in the launcher only: a caller
# that keeps a reference to the rasterizer's activated-parameter gradients (retain_grad, hooks) would see them overwritten.  An
# incoming gradient is written over ONLY if it lies in the rasterizer's own gradient allocation of the latest backward
# (diff_gaussian_rasterization._backend.is_in_grad_bucket): any other tensor -- a gradient autograd shares between nodes (fan-out of
# an add, a hook's copy) -- gets a fresh output, as the autograd contract demands.
INPLACE_GRAD = False


def _out_like(g, ref):
    """where the raw-parameter gradient goes: over the incoming gradient under INPLACE_GRAD (if that is the rasterizer's own gradient
    segment), else a fresh tensor"""
    if INPLACE_GRAD and g.is_contiguous() and g.dtype == torch.float32 and g.device == ref.device and g.numel() == ref.numel():
        from diff_gaussian_rasterization._backend import is_in_grad_bucket
        if is_in_grad_bucket(g):
            return g.view(ref.shape)
    return torch.empty_like(ref)
for _name, _n in (("gof_act_scaling", 3), ("gof_act_scaling_backward", 4), ("gof_act_opacity", 4), ("gof_act_opacity_backward", 6),
                  ("gof_act_rotation", 2), ("gof_act_rotation_backward", 3)):
    getattr(lib, _name).restype = C.c_int
    getattr(lib, _name).argtypes = [_i64] + [_vp] * _n + [_vp]


def _f3(filter_3D, n, like):
    f = B._need_cuda_f32(filter_3D.detach(), "filter_3D")
    if f.numel() != n or f.device != like.device:
        raise RuntimeError("filter_3D must hold one value per Gaussian (%d) on %s" % (n, like.device))
    return f.reshape(n)


class _Scaling(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_scaling, filter_3D):
        rs = B._need_cuda_f32(raw_scaling, "_scaling")
        n = int(rs.shape[0])
        f = _f3(filter_3D, n, rs)
        out = torch.empty_like(rs)
        with torch.cuda.device(rs.device):
            B._check(lib.gof_act_scaling(n, rs.data_ptr(), f.data_ptr(), out.data_ptr(), B._stream()))
        ctx.save_for_backward(rs, f)
        return out

    @staticmethod
    def backward(ctx, g):
        rs, f = ctx.saved_tensors
        g = g.contiguous()
        grs = _out_like(g, rs)
        with torch.cuda.device(rs.device):
            B._check(lib.gof_act_scaling_backward(int(rs.shape[0]), rs.data_ptr(), f.data_ptr(), g.data_ptr(), grs.data_ptr(), B._stream()))
        return grs, None


class _Opacity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_opacity, raw_scaling, filter_3D):
        ro = B._need_cuda_f32(raw_opacity, "_opacity")
        rs = B._need_cuda_f32(raw_scaling, "_scaling")
        n = int(rs.shape[0])
        f = _f3(filter_3D, n, rs)
        out = torch.empty_like(ro)
        with torch.cuda.device(rs.device):
            B._check(lib.gof_act_opacity(n, ro.data_ptr(), rs.data_ptr(), f.data_ptr(), out.data_ptr(), B._stream()))
        ctx.save_for_backward(ro, rs, f)
        return out

    @staticmethod
    def backward(ctx, g):
        ro, rs, f = ctx.saved_tensors
        g = g.contiguous()
        gro, grs = _out_like(g, ro), torch.empty_like(rs)
        with torch.cuda.device(rs.device):
            B._check(lib.gof_act_opacity_backward(int(rs.shape[0]), ro.data_ptr(), rs.data_ptr(), f.data_ptr(), g.data_ptr(), gro.data_ptr(),
                                                  grs.data_ptr(), B._stream()))
        return gro, grs, None


class _ScalingOpacity(torch.autograd.Function):
    """get_opacity_with_3D_filter and get_scaling_with_3D_filter as one node (INPLACE_GRAD): same kernels, same values; the backward
    leaves d/d_opacity over the incoming opacity gradient and d/d_scaling (both paths summed: the scaling path first, then the
    opacity path, the order autograd adds them in on the separate-node path) over the incoming scaling gradient."""

    @staticmethod
    def forward(ctx, raw_opacity, raw_scaling, filter_3D):
        ro = B._need_cuda_f32(raw_opacity, "_opacity")
        rs = B._need_cuda_f32(raw_scaling, "_scaling")
        n = int(rs.shape[0])
        f = _f3(filter_3D, n, rs)
        op, sc = torch.empty_like(ro), torch.empty_like(rs)
        with torch.cuda.device(rs.device):
            B._check(lib.gof_act_opacity(n, ro.data_ptr(), rs.data_ptr(), f.data_ptr(), op.data_ptr(), B._stream()))
            B._check(lib.gof_act_scaling(n, rs.data_ptr(), f.data_ptr(), sc.data_ptr(), B._stream()))
        ctx.save_for_backward(ro, rs, f)
        return op, sc

    @staticmethod
    def backward(ctx, g_op, g_sc):
        ro, rs, f = ctx.saved_tensors
        n = int(rs.shape[0])
        gro = grs = None
        global _backward_epoch
        _backward_epoch += 1          # a parked output of an older node must not be handed out any more (_fused_take)
        with torch.cuda.device(rs.device):
            if g_sc is not None:
                g_sc = g_sc.contiguous()
                grs = _out_like(g_sc, rs)
                B._check(lib.gof_act_scaling_backward(n, rs.data_ptr(), f.data_ptr(), g_sc.data_ptr(), grs.data_ptr(), B._stream()))
            if g_op is not None:
                g_op = g_op.contiguous()
                gro, grs_op = _out_like(g_op, ro), torch.empty_like(rs)
                B._check(lib.gof_act_opacity_backward(n, ro.data_ptr(), rs.data_ptr(), f.data_ptr(), g_op.data_ptr(), gro.data_ptr(),
                                                      grs_op.data_ptr(), B._stream()))
                grs = grs_op if grs is None else grs.add_(grs_op)
        return gro, grs, None


_backward_epoch = 0


def _fused_take(model, which):
    """`which` (0 = opacity, 1 = scaling) of `model` from a _ScalingOpacity node shared with the OTHER property: whichever is read
    first computes both and parks the other output, which the other property then takes -- once, so that every node is used for
    exactly one (opacity, scaling) pair (render() reads opacity at gaussian_renderer/__init__.py:60 and scaling at :70; a property
    read twice in a row simply starts a new node).  The key carries the count of backwards run so far: a parked output whose node has
    been through a backward (its graph freed) is never handed out."""
    key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (model._opacity, model._scaling, model.filter_3D)) + (torch.is_grad_enabled(), _backward_epoch)
    parked = getattr(model, "_gof_act_parked", None)
    model._gof_act_parked = None
    if parked is not None and parked[0] == key and parked[1] == which:
        return parked[2]
    pair = _ScalingOpacity.apply(model._opacity, model._scaling, model.filter_3D)
    model._gof_act_parked = (key, 1 - which, pair[1 - which])
    return pair[which]


class _Rotation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_rotation):
        rr = B._need_cuda_f32(raw_rotation, "_rotation")
        if rr.dim() != 2 or rr.shape[1] != 4:
            raise RuntimeError("_rotation must have dimensions (num_points, 4)")
        out = torch.empty_like(rr)
        with torch.cuda.device(rr.device):
            B._check(lib.gof_act_rotation(int(rr.shape[0]), rr.data_ptr(), out.data_ptr(), B._stream()))
        ctx.save_for_backward(rr)
        return out

    @staticmethod
    def backward(ctx, g):
        (rr,) = ctx.saved_tensors
        g = g.contiguous()
        grr = _out_like(g, rr)
        with torch.cuda.device(rr.device):
            B._check(lib.gof_act_rotation_backward(int(rr.shape[0]), rr.data_ptr(), g.data_ptr(), grr.data_ptr(), B._stream()))
        return grr


def scaling_with_3D_filter(raw_scaling, filter_3D):
    return _Scaling.apply(raw_scaling, filter_3D)


def opacity_with_3D_filter(raw_opacity, raw_scaling, filter_3D):
    return _Opacity.apply(raw_opacity, raw_scaling, filter_3D)


def rotation(raw_rotation):
    return _Rotation.apply(raw_rotation)


# property bodies for the reference's GaussianModel (self._scaling, self._opacity, self._rotation, self.filter_3D)
def get_scaling_with_3D_filter(self):
    if INPLACE_GRAD:
        return _fused_take(self, 1)
    return scaling_with_3D_filter(self._scaling, self.filter_3D)


def get_opacity_with_3D_filter(self):
    if INPLACE_GRAD:
        return _fused_take(self, 0)
    return opacity_with_3D_filter(self._opacity, self._scaling, self.filter_3D)


def get_rotation(self):
    return rotation(self._rotation)
