"""Mirror of the reference's utils/depth_utils.py: ``depths_to_points`` (:6-21) and ``depth_to_normal`` (:24-35),
one HIP launch forward and one backward instead of ~25 torch kernels (meshgrid, two matmuls, slicing, cross,
normalize and their autograd)."""
import math

import torch

from . import _backend as B


class _DepthToNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_hw, wvt, fx, fy):
        normals, points = B.depth_to_normal_forward(depth_hw, wvt, fx, fy)
        ctx.save_for_backward(depth_hw, wvt)
        ctx.fx, ctx.fy = fx, fy
        return normals, points

    @staticmethod
    def backward(ctx, g_normals, g_points):
        depth_hw, wvt = ctx.saved_tensors
        if g_normals is None:
            g_normals = torch.zeros(depth_hw.shape + (3,), dtype=torch.float32, device=depth_hw.device)
        g_normals = g_normals.contiguous()
        if g_points is not None:
            g_points = g_points.contiguous()
        return B.depth_to_normal_backward(depth_hw, wvt, ctx.fx, ctx.fy, g_normals, g_points), None, None, None


def _prepare(view, depth):
    W, H = int(view.image_width), int(view.image_height)
    if depth.numel() != W * H:
        raise RuntimeError("shape '[%d, 1]' is invalid for a view of %d x %d pixels" % (depth.numel(), W, H))
    fx = W / (2 * math.tan(view.FoVx / 2.))                          # depth_utils.py:9-10
    fy = H / (2 * math.tan(view.FoVy / 2.))
    wvt = view.world_view_transform
    if getattr(wvt, "requires_grad", False):
        raise NotImplementedError("depth_to_normal: gradient w.r.t. the camera pose is not implemented")
    wvt = B._need_cuda_f32(wvt, "view.world_view_transform")
    d = B._need_cuda_f32(depth, "depth").reshape(H, W)
    return d, wvt, float(fx), float(fy)


def depths_to_points(view, depthmap):
    """utils/depth_utils.py:6-21 -> (H*W, 3) world-space points."""
    d, wvt, fx, fy = _prepare(view, depthmap)
    _, points = _DepthToNormal.apply(d, wvt, fx, fy)
    return points.reshape(-1, 3)


def depth_to_normal(view, depth):
    """utils/depth_utils.py:24-35 -> (normals (H,W,3) with a zero 1-pixel border, points (H,W,3))."""
    d, wvt, fx, fy = _prepare(view, depth)
    normals, points = _DepthToNormal.apply(d, wvt, fx, fy)
    shape = tuple(depth.shape[1:]) + (3,)
    return normals.reshape(shape), points.reshape(shape)
