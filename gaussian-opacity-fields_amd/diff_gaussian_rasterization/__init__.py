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
        return _RasterizeGaussiansSplitSH.apply(means3D, means2D, sh[0], sh[1], opacities, scales, rotations, cov3Ds_precomp,
                                                view2gaussian_precomp, raster_settings)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     view2gaussian_precomp, raster_settings)


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
