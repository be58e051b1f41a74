"""MI355X-native ``diff_gaussian_rasterization``: the Python surface of the reference
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py) --
``GaussianRasterizationSettings``, ``GaussianRasterizer`` (forward / integrate / markVisible),
``rasterize_gaussians`` -- on top of hand-written gfx950 kernels (libgof_hip.so).

``gaussian_renderer.render()`` / ``integrate()`` of the reference import exactly these names
(gaussian_renderer/__init__.py:14) and run unchanged against this package.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _backend as _C
from ._backend import integrate_view_key, integrate_view_cache, IntegrateViewCache, integrate_min_into   # noqa: F401  (mesh-extraction driver fusion, NEW)


class GaussianRasterizationSettings(NamedTuple):
    # field order and names as the reference (diff_gaussian_rasterization/__init__.py:167-181)
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    subpixel_offset: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """CPU deep copy of an argument tuple for the debug dumps (reference :18-20)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_with_snapshot(fn, args, debug, dump_name, what):
    """pipe.debug behaviour of the reference (:89-96, :141-148): on failure dump the inputs and re-raise."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print("\nAn error occured in %s. Please forward %s for debugging." % (what, dump_name))
        raise


def _backward_with_mask_pool_redo(rs, args_of, fwd_args, geomBuffer, num_rendered, binningBuffer, imgBuffer):
    """The native backward; if the frame's forward ran with a contributor-mask pool that turned out too small (learnt capacity,
    _backend.MaskPoolTooSmall -- raised before anything is launched), its forward is repeated here with a full pool -- same inputs,
    hence the same image, lists and masks, now complete -- and the backward runs on those workspaces."""
    try:
        return _call_with_snapshot(_C.rasterize_gaussians_backward, args_of(geomBuffer, num_rendered, binningBuffer, imgBuffer),
                                   rs.debug, "snapshot_bw.dump", "backward")
    except _C.MaskPoolTooSmall:
        _C._stats["mask_pool_redone_frames"] += 1
        num_rendered, _color, _radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*fwd_args, fused=False)
        return _call_with_snapshot(_C.rasterize_gaussians_backward, args_of(geomBuffer, num_rendered, binningBuffer, imgBuffer),
                                   rs.debug, "snapshot_bw.dump", "backward")


# ---- the image's channel slices without autograd's per-slice zero-fill + add -------------------------------------------------------
# train.py takes four slices of the (9, H, W) image -- rendering[:3], [3:6], [6], [8] (train.py:149-172) -- and autograd's SliceBackward
# turns the gradient of EACH into a zero-filled (9, H, W) tensor with the slice copied in, then adds the four: four fills and three adds
# of 61 MB tensors at 1600x1063, ~0.18 ms of a 3.5 ms iteration (profiles/r05_full_loop_kernel_stats.md).  The image is therefore
# returned as a tensor subclass whose channel slices (an int or a slice on dimension 0, the other dimensions whole) are taken by a
# small autograd node of this package: its backward only REMEMBERS the slice's gradient; the first one to run hands autograd ONE
# zero-filled (9, H, W) buffer, the others nothing, and the rasterizer's backward copies the remembered gradients into whatever tensor
# autograd delivers (that buffer, or its sum with the gradient of a use of the whole image) before it reads it.  Same numbers (sums of
# the same terms; a channel nobody used is zero), any other operation on the image behaves as on a plain tensor and returns plain
# tensors.  Not covered: hooks / retain_grad on the image itself see the buffer before the slices are in it.  GOF_PLAIN_IMAGE=1
# returns the plain tensor.
import os as _os

_SLAB_IMAGE = _os.environ.get("GOF_PLAIN_IMAGE", "0") != "1"
_slice_hook = None          # train_epilogue/deferred.py: (slice, image, lo, hi, squeeze) -> what the script receives for a whole-channel slice


class _GradSlab:
    """per forward: the gradients of the image's channel slices on their way to the rasterizer's backward"""
    __slots__ = ("pending", "buf")

    def __init__(self):
        self.pending, self.buf = [], None

    def deliver(self, grad):
        """called by the rasterizer's backward with the gradient autograd delivers for the image: the remembered slices go in (in place:
        `grad` is this slab's own buffer, or a sum autograd formed for this node alone)"""
        if not self.pending:
            return grad
        own = self.buf is not None and grad.data_ptr() == self.buf.data_ptr() and grad.shape == self.buf.shape
        written = []
        for lo, hi, squeeze, g in self.pending:
            dst = grad[lo] if squeeze else grad[lo:hi]
            if own and all(hi <= a or b <= lo for a, b in written):
                dst.copy_(g)                    # the buffer is still zero there
            else:
                dst.add_(g)
            written.append((lo, hi))
        self.pending, self.buf = [], None
        return grad


class _ChannelSlice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, lo, hi, squeeze, slab):
        ctx.slab, ctx.lo, ctx.hi, ctx.squeeze, ctx.image_shape = slab, lo, hi, squeeze, image.shape
        ctx.set_materialize_grads(False)
        return image[lo] if squeeze else image[lo:hi]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None
        slab = ctx.slab
        first = slab.buf is None
        if first:
            slab.buf = torch.zeros(ctx.image_shape, dtype=g.dtype, device=g.device)
        slab.pending.append((ctx.lo, ctx.hi, ctx.squeeze, g))
        return (slab.buf if first else None), None, None, None, None


def _channel_range(index, channels):
    """(lo, hi, squeeze) if `index` selects whole channels of a (C, H, W) image -- an int or a step-1 slice on dimension 0, every other
    dimension taken whole -- else None"""
    if isinstance(index, tuple):
        if len(index) == 0 or len(index) > 3 or any(not (isinstance(r, slice) and r == slice(None)) and r is not Ellipsis for r in index[1:]):
            return None
        index = index[0]
    if isinstance(index, bool):
        return None
    if isinstance(index, int):
        i = index + channels if index < 0 else index
        return (i, i + 1, True) if 0 <= i < channels else None
    if isinstance(index, slice) and index.step in (None, 1):
        lo, hi, _ = index.indices(channels)
        return (lo, hi, False) if lo < hi else None
    return None


class RenderedImage(torch.Tensor):
    """The (9, H, W) image of a differentiable rasterizer call: a plain tensor in every respect but one -- see the comment above."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if (func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], RenderedImage) and torch.is_grad_enabled()
                and args[0].requires_grad and getattr(args[0], "_gof_slab", None) is not None and args[0].dim() == 3):
            r = _channel_range(args[1], args[0].shape[0])
            if r is not None:
                with torch._C.DisableTorchFunctionSubclass():
                    out = _ChannelSlice.apply(args[0].as_subclass(torch.Tensor), r[0], r[1], r[2], args[0]._gof_slab)
                return out if _slice_hook is None else _slice_hook(out, args[0], r[0], r[1], r[2])
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return out.as_subclass(torch.Tensor) if isinstance(out, RenderedImage) else out


def _as_rendered_image(color, slab):
    if not _SLAB_IMAGE or slab is None or not color.requires_grad:
        return color
    with torch._C.DisableTorchFunctionSubclass():
        img = color.as_subclass(RenderedImage)
    img._gof_slab = slab
    return img


def _view_args(rs, means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, sh):
    """Argument order of the native forward/integrate entry points after the leading (bg[, points3D])."""
    return (means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, view2gaussian_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.subpixel_offset,
            rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                view2gaussian_precomp, raster_settings):
        rs = raster_settings
        args = (rs.bg,) + _view_args(rs, means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     view2gaussian_precomp, sh)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _call_with_snapshot(
            _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.slab = _GradSlab() if _SLAB_IMAGE else None
        # (opacities: not an input of the reference's backward; kept -- a reference, no copy -- for the one case in which the frame's
        # forward has to be repeated before its backward: _backward_with_mask_pool_redo)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer, opacities)
        ctx.mark_non_differentiable(radii)
        # (autograd would otherwise hand the backward a zeros_like(radii) for the non-differentiable output: a 4 MB fill per step at 1M Gaussians)
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        if grad_out_color is None:          # (set_materialize_grads(False): the image took no part in the loss)
            return (None,) * 10
        if ctx.slab is not None:
            grad_out_color = ctx.slab.deliver(grad_out_color)      # the image's channel slices (RenderedImage)
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, radii, sh,
         geomBuffer, binningBuffer, imgBuffer, opacities) = ctx.saved_tensors

        def args_of(geom, num_rendered, binning, img):
            return (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    view2gaussian_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size,
                    rs.subpixel_offset, grad_out_color, sh, rs.sh_degree, rs.campos, geom, num_rendered, binning, img, rs.debug)
        fwd_args = (rs.bg,) + _view_args(rs, means3D, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, sh)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_view2gaussian_precomp) = _backward_with_mask_pool_redo(
            rs, args_of, fwd_args, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer)
        # one gradient per forward input, in the forward's order (reference :152-165)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, grad_view2gaussian_precomp, None)


class SplitSH:
    """The SH coefficients as the reference STORES them -- ``_features_dc`` (P,1,3) and ``_features_rest`` (P,15,3),
    scene/gaussian_model.py:351-352 -- handed to ``GaussianRasterizer`` as ``shs=SplitSH(dc, rest)`` instead of their
    concatenation (``GaussianModel.get_features``, gaussian_model.py:173-176: 192 B per Gaussian copied forward and its gradient
    split back every iteration, 0.22 ms at 1M Gaussians).  The kernels read / write the two tensors directly
    (``GofRasterArgs.shs_rest``); results are bit-identical to passing the concatenation.  NEW, not part of the reference API;
    launch/run_reference_script.py rebinds ``get_features`` to return one.  Any other use of the object (``.shape``,
    ``.transpose(...)``: gaussian_renderer/__init__.py:84-85 with ``pipe.convert_SHs_python``) sees the concatenated tensor."""

    def __init__(self, features_dc, features_rest):
        self.dc, self.rest = features_dc, features_rest

    def cat(self):
        return torch.cat((self.dc, self.rest), dim=1)

    def native(self):
        return self.dc.dim() == 3 and self.rest.dim() == 3 and self.dc.shape[1:] == (1, 3) and self.rest.shape[1:] == (15, 3)

    def __getattr__(self, name):              # only reached for attributes this class does not define
        return getattr(self.cat(), name)


class _RasterizeGaussiansSplitSH(torch.autograd.Function):
    """_RasterizeGaussians with the SH input as two tensors (SplitSH); same forward / backward entry points."""
    @staticmethod
    def forward(ctx, means3D, means2D, sh_dc, sh_rest, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, raster_settings):
        rs = raster_settings
        empty = torch.Tensor([])
        args = (rs.bg,) + _view_args(rs, means3D, empty, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, (sh_dc, sh_rest))
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _call_with_snapshot(
            _C.rasterize_gaussians, args, rs.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.slab = _GradSlab() if _SLAB_IMAGE else None
        ctx.save_for_backward(means3D, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, radii, sh_dc, sh_rest,
                              geomBuffer, binningBuffer, imgBuffer, opacities)
        ctx.mark_non_differentiable(radii)
        # (autograd would otherwise hand the backward a zeros_like(radii) for the non-differentiable output: a 4 MB fill per step at 1M Gaussians)
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        if grad_out_color is None:          # (set_materialize_grads(False): the image took no part in the loss)
            return (None,) * 10
        if ctx.slab is not None:
            grad_out_color = ctx.slab.deliver(grad_out_color)      # the image's channel slices (RenderedImage)
        rs = ctx.raster_settings
        (means3D, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, radii, sh_dc, sh_rest, geomBuffer, binningBuffer,
         imgBuffer, opacities) = ctx.saved_tensors
        empty = torch.Tensor([])

        def args_of(geom, num_rendered, binning, img):
            return (rs.bg, means3D, radii, empty, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                    view2gaussian_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size,
                    rs.subpixel_offset, grad_out_color, (sh_dc, sh_rest), rs.sh_degree, rs.campos, geom, num_rendered, binning, img, rs.debug)
        fwd_args = (rs.bg,) + _view_args(rs, means3D, empty, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, (sh_dc, sh_rest))
        (grad_means2D, _gc, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales, grad_rotations,
         grad_view2gaussian_precomp) = _backward_with_mask_pool_redo(rs, args_of, fwd_args, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer)
        return (grad_means3D, grad_means2D, grad_sh[0], grad_sh[1], grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, grad_view2gaussian_precomp, None)


def _split_or_cat(shs):
    """SplitSH -> (dc, rest) when the kernels take it as it is (SH degree-3 storage), else its concatenation."""
    if isinstance(shs, SplitSH):
        return (shs.dc, shs.rest) if shs.native() else shs.cat()
    return shs


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        view2gaussian_precomp, raster_settings):
    if isinstance(sh, tuple):
        color, radii = _RasterizeGaussiansSplitSH.apply(means3D, means2D, sh[0], sh[1], opacities, scales, rotations, cov3Ds_precomp,
                                                        view2gaussian_precomp, raster_settings)
    else:
        color, radii = _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                                 view2gaussian_precomp, raster_settings)
    fn = color.grad_fn
    return _as_rendered_image(color, getattr(fn, "slab", None) if fn is not None else None), radii


def _normalise_optionals(shs, colors_precomp, scales, rotations, cov3D_precomp, view2gaussian_precomp):
    """Input contract of GaussianRasterizer.forward/integrate (reference :203-223): exactly one colour source,
    exactly one covariance source; absent inputs become empty tensors (NULL for the native side)."""
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    has_sr = scales is not None or rotations is not None
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    empty = torch.Tensor([])
    fill = lambda t: empty if t is None else t   # noqa: E731
    return fill(shs), fill(colors_precomp), fill(scales), fill(rotations), fill(cov3D_precomp), fill(view2gaussian_precomp)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the points that pass the near-plane test of the camera (reference :188-197)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, view2gaussian_precomp=None):
        shs, colors_precomp, scales, rotations, cov3D_precomp, view2gaussian_precomp = _normalise_optionals(
            _split_or_cat(shs), colors_precomp, scales, rotations, cov3D_precomp, view2gaussian_precomp)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   view2gaussian_precomp, self.raster_settings)

    def integrate(self, points3D, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                  cov3D_precomp=None, view2gaussian_precomp=None):
        """Opacity-field query: no autograd (reference :239-305).  Returns
        ``(color, alpha_integrated, color_integrated, radii)``."""
        shs, colors_precomp, scales, rotations, cov3D_precomp, view2gaussian_precomp = _normalise_optionals(
            _split_or_cat(shs), colors_precomp, scales, rotations, cov3D_precomp, view2gaussian_precomp)
        rs = self.raster_settings
        args = (rs.bg, points3D) + _view_args(rs, means3D, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                              view2gaussian_precomp, shs)
        (num_rendered, color, alpha_integrated, color_integrated, radii, geomBuffer, binningBuffer, imgBuffer) = \
            _call_with_snapshot(_C.integrate_gaussians_to_points, args, rs.debug, "snapshot_fw.dump", "forward")
        return color, alpha_integrated, color_integrated, radii
