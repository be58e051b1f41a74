"""ctypes binding of the training-epilogue entry points of libgof_hip.so (include/gof_train_hip.h).
Shares the library handle of the rasterizer backend; no fallback (import fails if the library is missing)."""
import ctypes as C

import torch

from diff_gaussian_rasterization import _backend as _B

lib = _B.lib
_check = _B._check
_stream = _B._stream

ADAM_MAX_TENSORS = 16
SSIM_WINDOW = 11


class GofAdamTensor(C.Structure):
    """Mirror of ``GofAdamTensor`` in include/gof_train_hip.h."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_uint64), ("step_size", C.c_float), ("bias_correction2_sqrt", C.c_float)]


def _declare():
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int32, C.c_float
    W11 = C.POINTER(C.c_float)
    lib.gof_ssim_scratch_bytes.restype = sz
    lib.gof_ssim_scratch_bytes.argtypes = [i32, i32, i32]
    lib.gof_ssim_forward.argtypes = [i32, i32, i32, vp, vp, W11, vp, vp, vp, sz, vp]
    lib.gof_ssim_backward.argtypes = [i32, i32, i32, vp, vp, W11, vp, vp, vp, vp]
    lib.gof_depth_to_normal.argtypes = [i32, i32, vp, vp, f32, f32, vp, vp, vp]
    lib.gof_depth_to_normal_backward.argtypes = [i32, i32, vp, vp, f32, f32, vp, vp, vp, vp]
    lib.gof_adam_step.argtypes = [i32, C.POINTER(GofAdamTensor), C.c_double, C.c_double, C.c_double, vp]
    lib.gof_l1_scratch_bytes.restype = sz
    lib.gof_l1_scratch_bytes.argtypes = [C.c_uint64]
    lib.gof_l1_forward.argtypes = [C.c_uint64, vp, vp, vp, vp, sz, vp]
    lib.gof_l1_backward.argtypes = [C.c_uint64, vp, vp, vp, vp, vp]
    lib.gof_l1_forward.restype = lib.gof_l1_backward.restype = C.c_int
    lib.gof_train_loss_scratch_bytes.restype = sz
    lib.gof_train_loss_scratch_bytes.argtypes = [i32, i32]
    lib.gof_train_loss.argtypes = [i32, i32, vp, vp, W11, vp, f32, f32, C.c_double, C.c_double, C.c_double, vp, vp, vp, sz, vp]
    lib.gof_train_loss.restype = C.c_int
    lib.gof_rot3_apply.argtypes = [C.c_int64, vp, i32, i32, i32, vp, vp, vp]
    lib.gof_rot3_apply.restype = C.c_int
    for n in ("gof_ssim_forward", "gof_ssim_backward", "gof_depth_to_normal", "gof_depth_to_normal_backward", "gof_adam_step"):
        getattr(lib, n).restype = C.c_int


_declare()


def _need_cuda_f32(t, what):
    if t.device.type != "cuda":
        raise RuntimeError("%s must be on a ROCm device (got %s); the gfx950 backend has no CPU path" % (what, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s)" % (what, t.dtype))
    return t.contiguous()


def _same_device(*tensors):
    """The device all operands live on (None entries skipped); a launch is enqueued on THAT device's current stream, so operands
    spread over several GPUs are refused (the library has no hipSetDevice of its own)."""
    ts = [t for t in tensors if t is not None]
    dev = ts[0].device
    for t in ts:
        if t.device != dev:
            raise RuntimeError("all operands must be on one ROCm device (got %s and %s)" % (dev, t.device))
    return torch.cuda.device(dev)


def window_taps(window_size=SSIM_WINDOW, sigma=1.5):
    """The 11 fp32 taps exactly as utils/loss_utils.py:22-24 builds them (python doubles -> fp32 tensor -> / fp32 sum)."""
    from math import exp
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = g / g.sum()
    return (C.c_float * window_size)(*[float(v) for v in g])


def ssim_forward(img1, img2, taps, want_grad):
    planes = img1.numel() // (img1.shape[-1] * img1.shape[-2])
    H, W = int(img1.shape[-2]), int(img1.shape[-1])
    dev = img1.device
    sums = torch.empty(planes, dtype=torch.float32, device=dev)
    dmaps = torch.empty((3, planes, H, W), dtype=torch.float32, device=dev) if want_grad else None
    nb = lib.gof_ssim_scratch_bytes(planes, W, H)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    with _same_device(img1, img2):
        _check(lib.gof_ssim_forward(planes, W, H, img1.data_ptr(), img2.data_ptr(), taps, sums.data_ptr(),
                                    dmaps.data_ptr() if want_grad else None, scratch.data_ptr(), nb, _stream()))
    return sums, dmaps


def ssim_backward(img1, img2, taps, dmaps, plane_scale):
    planes = dmaps.shape[1]
    H, W = int(img1.shape[-2]), int(img1.shape[-1])
    out = torch.empty_like(img1)
    with _same_device(img1, img2, dmaps, plane_scale):
        _check(lib.gof_ssim_backward(planes, W, H, img1.data_ptr(), img2.data_ptr(), taps, dmaps.data_ptr(),
                                     plane_scale.data_ptr(), out.data_ptr(), _stream()))
    return out


def depth_to_normal_forward(depth_hw, wvt, fx, fy):
    H, W = int(depth_hw.shape[0]), int(depth_hw.shape[1])
    normals = torch.empty((H, W, 3), dtype=torch.float32, device=depth_hw.device)
    points = torch.empty((H, W, 3), dtype=torch.float32, device=depth_hw.device)
    with _same_device(depth_hw, wvt):
        _check(lib.gof_depth_to_normal(W, H, depth_hw.data_ptr(), wvt.data_ptr(), fx, fy, normals.data_ptr(), points.data_ptr(), _stream()))
    return normals, points


def depth_to_normal_backward(depth_hw, wvt, fx, fy, g_normals, g_points):
    H, W = int(depth_hw.shape[0]), int(depth_hw.shape[1])
    out = torch.empty((H, W), dtype=torch.float32, device=depth_hw.device)
    with _same_device(depth_hw, wvt, g_normals, g_points):
        _check(lib.gof_depth_to_normal_backward(W, H, depth_hw.data_ptr(), wvt.data_ptr(), fx, fy, g_normals.data_ptr(),
                                                g_points.data_ptr() if g_points is not None else None, out.data_ptr(), _stream()))
    return out


def rot3_apply(m33, x, transpose=False):
    """Y [3,N] = A X, A = the 3x3 (possibly strided) float32 view m33, or its transpose (include/gof_train_hip.h: gof_rot3_apply)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    with _same_device(m33, x):
        _check(lib.gof_rot3_apply(int(x.shape[1]), m33.data_ptr(), int(m33.stride(0)), int(m33.stride(1)), 1 if transpose else 0,
                                  x.data_ptr(), out.data_ptr(), _stream()))
    return out


def l1_forward(a, b):
    n = a.numel()
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    nb = lib.gof_l1_scratch_bytes(n)
    scratch = torch.empty(nb, dtype=torch.uint8, device=a.device)
    with _same_device(a, b):
        _check(lib.gof_l1_forward(n, a.data_ptr(), b.data_ptr(), out.data_ptr(), scratch.data_ptr(), nb, _stream()))
    return out.reshape(())


def l1_backward(a, b, grad_out):
    out = torch.empty_like(a)
    with _same_device(a, b, grad_out):
        _check(lib.gof_l1_backward(a.numel(), a.data_ptr(), b.data_ptr(), grad_out.data_ptr(), out.data_ptr(), _stream()))
    return out


def train_loss(rendering, gt_image, taps, wvt, fx, fy, lambda_dssim, lambda_depth_normal, lambda_distortion, want_grad):
    """gof_train_loss: (terms[6] = loss, Ll1, ssim, rgb_loss, depth_normal_loss, distortion_loss; d loss / d rendering or None)."""
    H, W = int(rendering.shape[-2]), int(rendering.shape[-1])
    dev = rendering.device
    terms = torch.empty(6, dtype=torch.float32, device=dev)
    dL = torch.empty_like(rendering) if want_grad else None
    nb = lib.gof_train_loss_scratch_bytes(W, H)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    with _same_device(rendering, gt_image, wvt):
        _check(lib.gof_train_loss(W, H, rendering.data_ptr(), gt_image.data_ptr(), taps, wvt.data_ptr(), fx, fy, float(lambda_dssim),
                                  float(lambda_depth_normal), float(lambda_distortion), terms.data_ptr(),
                                  dL.data_ptr() if want_grad else None, scratch.data_ptr(), nb, _stream()))
    return terms, dL


def adam_step(entries, beta1, beta2, eps):
    """entries: list of (param, grad, exp_avg, exp_avg_sq, step_size, bias_correction2_sqrt)."""
    for i in range(0, len(entries), ADAM_MAX_TENSORS):
        chunk = entries[i:i + ADAM_MAX_TENSORS]
        arr = (GofAdamTensor * len(chunk))()
        for k, (p, g, m, v, step_size, bc2s) in enumerate(chunk):
            arr[k].param, arr[k].grad, arr[k].exp_avg, arr[k].exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            arr[k].n = p.numel()
            arr[k].step_size, arr[k].bias_correction2_sqrt = step_size, bc2s
        with _same_device(*[t for e in chunk for t in e[:4]]):
            _check(lib.gof_adam_step(len(chunk), arr, beta1, beta2, eps, _stream()))
