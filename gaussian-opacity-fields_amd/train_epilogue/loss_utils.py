"""Mirror of the reference's utils/loss_utils.py (same names, arguments and results) with SSIM on HIP kernels.

    reference                          here
    l1_loss           (:17-18)         one forward launch (+ a deterministic reduction) and one backward launch instead of
                                        sub / abs / mean and their autograd (6 launches)
    l2_loss           (:20-21)         unchanged torch one-liner (not used by train.py)
    ssim / _ssim      (:30-63)         one forward launch (+ a deterministic reduction) and one backward launch
                                        instead of 5 depthwise conv2d + ~15 elementwise kernels and their autograd
"""
import torch

from . import _backend as B


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return B.l1_forward(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = B.l1_backward(a, b, g.to(torch.float32).contiguous()) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        return (ga if ctx.needs_input_grad[0] else None), (-ga if ctx.needs_input_grad[1] else None)


def l1_loss(network_output, gt):
    """utils/loss_utils.py:17-18: torch.abs((network_output - gt)).mean().  Same-shape float32 tensors (what train.py:156,328 pass)
    take the two HIP launches; broadcasting / other dtypes take the reference's expression as torch ops on the same device.
    Like every mirror of this package there is no CPU path: host tensors raise."""
    if network_output.device.type != "cuda" or gt.device != network_output.device:
        raise RuntimeError("l1_loss: both tensors must be on the same ROCm device (got %s and %s); the gfx950 backend has no CPU path"
                           % (network_output.device, gt.device))
    if network_output.shape == gt.shape and network_output.dtype == gt.dtype == torch.float32 and network_output.numel() > 0:
        return _L1.apply(network_output.contiguous(), gt.contiguous())
    return torch.abs((network_output - gt)).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


_TAPS = {}


def _taps(window_size):
    if window_size != B.SSIM_WINDOW:
        raise NotImplementedError("ssim: the gfx950 kernel implements window_size=11 (the reference's default and only use), got %r" % (window_size,))
    if window_size not in _TAPS:
        _TAPS[window_size] = B.window_taps(window_size)
    return _TAPS[window_size]


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, window_size, size_average):
        taps = _taps(window_size)
        if img1.shape != img2.shape:
            raise RuntimeError("ssim: img1 and img2 must have the same shape, got %s and %s" % (tuple(img1.shape), tuple(img2.shape)))
        if img1.dim() not in (3, 4):
            raise RuntimeError("ssim: expected (C,H,W) or (B,C,H,W) input, got %d-D" % img1.dim())
        if not size_average and img1.dim() != 4:
            raise IndexError("Dimension out of range (ssim with size_average=False needs a batched (B,C,H,W) input)")
        x = B._need_cuda_f32(img1, "img1")
        y = B._need_cuda_f32(img2, "img2")
        want = ctx.needs_input_grad[0]
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("ssim: gradient w.r.t. img2 (the ground-truth image) is not implemented")
        sums, dmaps = B.ssim_forward(x, y, taps, want)
        ctx.size_average = size_average
        ctx.taps = taps
        ctx.per_plane = x.shape[-1] * x.shape[-2]
        ctx.channels = x.shape[-3]
        if want:
            ctx.save_for_backward(x, y, dmaps)
        if size_average:
            return sums.sum() / float(x.numel())                                  # loss_utils.py:60-61
        return sums.view(x.shape[0], ctx.channels).sum(1) / float(ctx.channels * ctx.per_plane)   # :62-63

    @staticmethod
    def backward(ctx, grad_out):
        x, y, dmaps = ctx.saved_tensors
        planes = dmaps.shape[1]
        if ctx.size_average:
            scale = (grad_out.to(torch.float32) / float(x.numel())).reshape(1).expand(planes).contiguous()
        else:
            scale = (grad_out.to(torch.float32) / float(ctx.channels * ctx.per_plane)).reshape(-1, 1).expand(-1, ctx.channels).contiguous().view(-1)
        return B.ssim_backward(x, y, ctx.taps, dmaps, scale), None, None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:30-41: mean SSIM with an 11x11 Gaussian window (sigma 1.5), zero padding, per-channel."""
    return _Ssim.apply(img1, img2, window_size, size_average)
