"""The loss of one training iteration of the reference (train.py:150-188) as one operator over the rasterizer's output.

train.py composes it inline -- l1_loss, ssim, a mean, depth_to_normal, F.normalize, a 3x3 matmul, a dot product, two more
means -- which costs ~60 torch launches and their autograd per iteration (1.0 ms of a 5.0 ms iteration at 1600x1063 once
the rasterizer runs on this backend).  ``training_loss`` evaluates the same expression with five HIP launches
(``gof_train_loss``, include/gof_train_hip.h).  It is NOT a drop-in for an unchanged train.py (the composition is inline
there, nothing to rebind by name); INTEGRATION.md shows the seven-line change that uses it.  The mirrors in loss_utils.py /
depth_utils.py remain the default for the unchanged script.
"""
import collections
import math

import torch

from . import _backend as B
from .loss_utils import _taps

TrainingLoss = collections.namedtuple("TrainingLoss", "loss Ll1 ssim rgb_loss depth_normal_loss distortion_loss")


class _TrainingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rendering, gt_image, wvt, fx, fy, lambda_dssim, lambda_depth_normal, lambda_distortion):
        terms, dL = B.train_loss(rendering, gt_image, _taps(B.SSIM_WINDOW), wvt, fx, fy, lambda_dssim, lambda_depth_normal,
                                 lambda_distortion, ctx.needs_input_grad[0])
        ctx.save_for_backward(dL)
        ctx.set_materialize_grads(False)                     # no zero-filled gradients for the five logging outputs
        outs = tuple(terms[i] for i in range(6))
        ctx.mark_non_differentiable(*outs[1:])               # the individual terms are for logging (train.py:239-247)
        return outs

    @staticmethod
    def backward(ctx, g_loss, *unused):
        (dL,) = ctx.saved_tensors
        return (dL * g_loss if (dL is not None and g_loss is not None) else None), None, None, None, None, None, None, None


def _checked(rendering, gt_image, viewpoint_cam, who):
    """the argument checks of training_loss -> (rendering, gt_image, world_view_transform, fx, fy) as the native call wants them"""
    if rendering.dim() != 3 or rendering.shape[0] != 9:
        raise RuntimeError("%s: rendering must be the rasterizer's (9,H,W) output, got %s" % (who, tuple(rendering.shape)))
    if gt_image.dim() != 3 or gt_image.shape[0] != 3 or gt_image.shape[1:] != rendering.shape[1:]:
        raise RuntimeError("%s: gt_image must be (3,%d,%d), got %s" % (who, rendering.shape[1], rendering.shape[2], tuple(gt_image.shape)))
    if gt_image.requires_grad:
        raise NotImplementedError("%s: gradient w.r.t. the ground-truth image is not implemented" % who)
    H, W = int(rendering.shape[1]), int(rendering.shape[2])
    if int(viewpoint_cam.image_width) != W or int(viewpoint_cam.image_height) != H:
        raise RuntimeError("%s: the camera is %dx%d, the rendering %dx%d" % (who, viewpoint_cam.image_width, viewpoint_cam.image_height, W, H))
    fx = W / (2 * math.tan(viewpoint_cam.FoVx / 2.))                 # depth_utils.py:9-10
    fy = H / (2 * math.tan(viewpoint_cam.FoVy / 2.))
    wvt = viewpoint_cam.world_view_transform
    if getattr(wvt, "requires_grad", False):
        raise NotImplementedError("%s: gradient w.r.t. the camera pose is not implemented" % who)
    r = B._need_cuda_f32(rendering, "rendering")
    g = B._need_cuda_f32(gt_image, "gt_image")
    w = B._need_cuda_f32(wvt, "viewpoint_cam.world_view_transform")
    return r, g, w, float(fx), float(fy)


def training_loss(rendering, gt_image, viewpoint_cam, lambda_dssim=0.2, lambda_depth_normal=0.0, lambda_distortion=0.0):
    """train.py:150-188 for ``rendering`` = render(...)["render"] (9,H,W) and the ground-truth image (3,H,W):

        loss = (1 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt))
               + lambda_depth_normal * depth_normal_loss + lambda_distortion * distortion_loss

    with the lambdas the caller has already gated by iteration (train.py:184-185).  Returns a ``TrainingLoss`` of 0-dim
    tensors; only ``loss`` carries a gradient (w.r.t. ``rendering``)."""
    r, g, w, fx, fy = _checked(rendering, gt_image, viewpoint_cam, "training_loss")
    return TrainingLoss(*_TrainingLoss.apply(r, g, w, fx, fy, float(lambda_dssim), float(lambda_depth_normal), float(lambda_distortion)))


def training_loss_gradient(rendering, gt_image, viewpoint_cam, lambda_dssim, lambda_depth_normal, lambda_distortion):
    """The same call WITHOUT an autograd node: (TrainingLoss of 0-dim value tensors, d loss / d rendering (9,H,W)).  For a caller that
    hands the gradient to ``torch.autograd.backward(rendering, dL)`` itself (deferred.py: the loss of an unchanged train.py) -- the
    node's backward would multiply the 9 H W gradient by an upstream gradient that is known to be 1."""
    r, g, w, fx, fy = _checked(rendering.detach(), gt_image, viewpoint_cam, "training_loss_gradient")
    terms, dL = B.train_loss(r, g, _taps(B.SSIM_WINDOW), w, fx, fy, float(lambda_dssim), float(lambda_depth_normal), float(lambda_distortion), True)
    return TrainingLoss(*(terms[i] for i in range(6))), dL
