"""CPU restatement of the training-epilogue functions of the reference (TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path).

Each function follows the reference file:line it cites and runs on CPU tensors, so that torch autograd of the
restatement is the gradient oracle.  Pinned against the reference's own Python (executed where it lies, on CPU) by
tests/golden/make_golden_train.py -> tests/golden/ref_train_epilogue_golden.npz (tests/test_train_epilogue_oracle.py).
"""
import math
from math import exp

import numpy as np
import torch
import torch.nn.functional as F


# ---- utils/loss_utils.py ------------------------------------------------------------------------------
def l1_loss(network_output, gt):                     # loss_utils.py:17-18
    return torch.abs((network_output - gt)).mean()


def gaussian_window_1d(window_size=11, sigma=1.5):   # loss_utils.py:22-24
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def ssim(img1, img2, window_size=11, size_average=True):
    """loss_utils.py:26-63 (create_window + ssim + _ssim)."""
    channel = img1.size(-3)
    w1 = gaussian_window_1d(window_size).unsqueeze(1)                      # :27
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)                   # :28
    window = w2.expand(channel, 1, window_size, window_size).contiguous().type_as(img1)   # :29, :39
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)              # :44-45
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2            # :47-49
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq   # :51-53
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2                                          # :54-55
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))   # :57
    if size_average:
        return ssim_map.mean()                                             # :60
    return ssim_map.mean(1).mean(1).mean(1)                                # :62


# ---- utils/depth_utils.py -----------------------------------------------------------------------------
def depths_to_points(world_view_transform, W, H, FoVx, FoVy, depthmap, dtype=torch.float32):
    """depth_utils.py:6-21 with the camera fields passed explicitly (the reference hard-codes .cuda()).
    dtype=torch.float64 (with float64 inputs) evaluates the same expressions in double: the accuracy yardstick of the
    GPU tests, not part of the restatement."""
    c2w = (world_view_transform.T).inverse()                               # :7
    fx = W / (2 * math.tan(FoVx / 2.))                                     # :9-10
    fy = H / (2 * math.tan(FoVy / 2.))
    intrins = torch.tensor([[fx, 0., W / 2.], [0., fy, H / 2.], [0., 0., 1.0]], dtype=torch.float64).to(dtype)   # :11-15 (.float())
    grid_x, grid_y = torch.meshgrid(torch.arange(W).to(dtype) + 0.5, torch.arange(H).to(dtype) + 0.5, indexing='xy')  # :16
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=-1).reshape(-1, 3)   # :17
    rays_d = points @ intrins.inverse().T @ c2w[:3, :3].T                  # :18
    rays_o = c2w[:3, 3]                                                    # :19
    return depthmap.reshape(-1, 1) * rays_d + rays_o                       # :20


def depth_to_normal(world_view_transform, W, H, FoVx, FoVy, depth, dtype=torch.float32):
    """depth_utils.py:24-35."""
    points = depths_to_points(world_view_transform, W, H, FoVx, FoVy, depth, dtype).reshape(*depth.shape[1:], 3)   # :29
    output = torch.zeros_like(points)                                      # :30
    dx = torch.cat([points[2:, 1:-1] - points[:-2, 1:-1]], dim=0)          # :31
    dy = torch.cat([points[1:-1, 2:] - points[1:-1, :-2]], dim=1)          # :32
    normal_map = F.normalize(torch.cross(dx, dy, dim=-1), dim=-1)          # :33
    output[1:-1, 1:-1, :] = normal_map                                     # :34
    return output, points


# ---- train.py: the loss of one iteration --------------------------------------------------------------
def training_loss(rendering, gt_image, world_view_transform, W, H, FoVx, FoVy, lambda_dssim, lambda_depth_normal, lambda_distortion,
                  dtype=torch.float32):
    """train.py:150-188 (without the decoupled-appearance branch, :158-159) on CPU tensors; the lambdas already gated by
    iteration (:184-185).  Returns (loss, Ll1, ssim, rgb_loss, depth_normal_loss, distortion_loss)."""
    image = rendering[:3, :, :]                                                           # :150
    Ll1 = l1_loss(image, gt_image)                                                        # :156
    ssim_value = ssim(image, gt_image)
    rgb_loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim_value)             # :161
    distortion_map = rendering[8, :, :]                                                   # :164
    distortion_loss = distortion_map.mean()                                               # :167
    depth = rendering[6, :, :]                                                            # :170
    depth_normal, _ = depth_to_normal(world_view_transform, W, H, FoVx, FoVy, depth[None, ...], dtype)   # :171
    depth_normal = depth_normal.permute(2, 0, 1)                                          # :172
    render_normal = rendering[3:6, :, :]                                                  # :174
    render_normal = F.normalize(render_normal, p=2, dim=0)                                # :175
    c2w = (world_view_transform.T).inverse()                                              # :177
    normal2 = c2w[:3, :3] @ render_normal.reshape(3, -1)                                  # :178
    render_normal_world = normal2.reshape(3, *render_normal.shape[1:])                    # :179
    normal_error = 1 - (render_normal_world * depth_normal).sum(dim=0)                    # :181
    depth_normal_loss = normal_error.mean()                                               # :182
    loss = rgb_loss + depth_normal_loss * lambda_depth_normal + distortion_loss * lambda_distortion   # :188
    return loss, Ll1, ssim_value, rgb_loss, depth_normal_loss, distortion_loss


# ---- scene/gaussian_model.py: compute_3D_filter ------------------------------------------------------------
def compute_3d_filter(xyz, cameras):
    """GaussianModel.compute_3D_filter (gaussian_model.py:262-311) on CPU tensors; cameras: objects with R, T (numpy), focal_x,
    focal_y, image_width, image_height.  Returns filter_3D (P,1)."""
    distance = torch.ones((xyz.shape[0])) * 100000.0                                      # :267
    valid_points = torch.zeros((xyz.shape[0]), dtype=torch.bool)                          # :268
    focal_length = 0.                                                                     # :271
    for camera in cameras:
        R = torch.tensor(camera.R, dtype=torch.float32)                                   # :275-276
        T = torch.tensor(camera.T, dtype=torch.float32)
        xyz_cam = xyz @ R + T[None, :]                                                    # :278
        valid_depth = xyz_cam[:, 2] > 0.2                                                 # :283
        x, y, z = xyz_cam[:, 0], xyz_cam[:, 1], xyz_cam[:, 2]
        z = torch.clamp(z, min=0.001)                                                     # :287
        x = x / z * camera.focal_x + camera.image_width / 2.0                             # :289-290
        y = y / z * camera.focal_y + camera.image_height / 2.0
        in_screen = torch.logical_and(torch.logical_and(x >= -0.15 * camera.image_width, x <= camera.image_width * 1.15),
                                      torch.logical_and(y >= -0.15 * camera.image_height, y <= 1.15 * camera.image_height))   # :295
        valid = torch.logical_and(valid_depth, in_screen)                                 # :298
        distance[valid] = torch.min(distance[valid], z[valid])                            # :301
        valid_points = torch.logical_or(valid_points, valid)                              # :302
        if focal_length < camera.focal_x:                                                 # :303-304
            focal_length = camera.focal_x
    distance[~valid_points] = distance[valid_points].max()                                # :306
    filter_3D = distance / focal_length * (0.2 ** 0.5)                                    # :310
    return filter_3D[..., None]


def add_densification_stats(accum, accum_abs, accum_abs_max, denom, grad, update_filter):
    """GaussianModel.add_densification_stats (gaussian_model.py:709-714) on CPU tensors, in place."""
    accum[update_filter] += torch.norm(grad[update_filter, :2], dim=-1, keepdim=True)                       # :710
    accum_abs[update_filter] += torch.norm(grad[update_filter, 2:], dim=-1, keepdim=True)                    # :712
    accum_abs_max[update_filter] = torch.max(accum_abs_max[update_filter], torch.norm(grad[update_filter, 2:], dim=-1, keepdim=True))   # :713
    denom[update_filter] += 1                                                                                # :714


# ---- scene/gaussian_model.py: the derived tensors render() reads ------------------------------------------------------------
def scaling_with_3D_filter(raw_scaling, filter_3D):
    scales = torch.exp(raw_scaling)                                                       # get_scaling, :152-154
    scales = torch.square(scales) + torch.square(filter_3D)                               # :160
    return torch.sqrt(scales)                                                             # :161


def opacity_with_3D_filter(raw_opacity, raw_scaling, filter_3D):
    opacity = torch.sigmoid(raw_opacity)                                                  # :184
    scales = torch.exp(raw_scaling)                                                       # :186
    scales_square = torch.square(scales)                                                  # :188
    det1 = scales_square.prod(dim=1)                                                      # :189
    scales_after_square = scales_square + torch.square(filter_3D)                         # :191
    det2 = scales_after_square.prod(dim=1)                                                # :192
    coef = torch.sqrt(det1 / det2)                                                        # :193
    return opacity * coef[..., None]                                                      # :194


def rotation(raw_rotation):
    return F.normalize(raw_rotation)                                                      # :166


# ---- torch/optim/adam.py (the optimizer scene/gaussian_model.py:360 builds) ----------------------------
def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """One step of torch.optim.Adam (no weight decay, no amsgrad) in fp32 numpy, operation order of
    torch/optim/adam.py::_single_tensor_adam.  `step` is the 1-based step count.  Returns (param, exp_avg, exp_avg_sq)."""
    f = np.float32
    p, g, m, v = (np.asarray(a, dtype=f) for a in (param, grad, exp_avg, exp_avg_sq))
    m = m + f(1 - beta1) * (g - m)                                         # exp_avg.lerp_(grad, 1 - beta1)
    v = v * f(beta2) + (f(1 - beta2) * g) * g                              # mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bias_correction1 = 1 - beta1 ** step
    bias_correction2 = 1 - beta2 ** step
    step_size = lr / bias_correction1
    bias_correction2_sqrt = bias_correction2 ** 0.5
    denom = np.sqrt(v) / f(bias_correction2_sqrt) + f(eps)                 # (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p + (f(-step_size) * m) / denom                                    # param.addcdiv_(exp_avg, denom, value=-step_size)
    return p.astype(f), m.astype(f), v.astype(f)
