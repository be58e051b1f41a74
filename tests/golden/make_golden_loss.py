"""Generates tests/golden/ref_train_loss_golden.npz by EXECUTING the reference's own Python where it lies (/root/reference; build
container only -- the GPU box reads the committed .npz): the loss of one training iteration, train.py:150-188, composed from the
reference's own utils.loss_utils.l1_loss / ssim and utils.depth_utils.depth_to_normal exactly as train.py composes them inline
(the statements between them are torch calls; this script issues the same calls in the same order), with autograd for the gradient
w.r.t. the rasterizer output.  `.cuda()` / `device='cuda'` are redirected to the CPU as in make_golden_train.py.
No reference code is copied."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

torch.Tensor.cuda = lambda self, *a, **k: self
_arange = torch.arange
torch.arange = lambda *a, **k: _arange(*a, **{kk: v for kk, v in k.items() if kk != "device"})

from utils.loss_utils import l1_loss, ssim                       # noqa: E402
from utils.depth_utils import depth_to_normal                    # noqa: E402


def rigid(seed):
    gg = torch.Generator().manual_seed(seed)
    q = torch.randn(4, generator=gg)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    R = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    t = torch.randn(3, generator=gg)
    M = torch.eye(4)
    M[:3, :3] = R
    M[:3, 3] = t
    return M.T.contiguous()          # world_view_transform is stored transposed (scene/cameras.py:56)


def synthetic_rendering(W, H, g):
    """A (9,H,W) tensor shaped like the rasterizer's output: colour, alpha-weighted normals (some exactly zero = empty pixels),
    positive depth (zero where empty), alpha, non-negative distortion."""
    r = torch.empty(9, H, W)
    r[0:3] = torch.rand(3, H, W, generator=g)
    r[3:6] = torch.randn(3, H, W, generator=g) * 0.6
    r[6] = 1.0 + 4.0 * torch.rand(H, W, generator=g)
    r[7] = torch.rand(H, W, generator=g)
    r[8] = torch.rand(H, W, generator=g) * 0.01
    empty = torch.rand(H, W, generator=g) < 0.05
    r[3:6, empty] = 0.0
    r[6, empty] = 0.0
    return r


out = {}
g = torch.Generator().manual_seed(4321)
cases = (("a", (37, 29, 0.9, 0.7, 11), (0.2, 0.05, 100.0)), ("b", (16, 16, 1.2, 1.2, 12), (0.2, 0.0, 0.0)),
         ("c", (45, 18, 0.5, 0.6, 13), (0.35, 0.05, 1000.0)))
for tag, (W, H, fovx, fovy, seed), (lambda_dssim, lambda_depth_normal, lambda_distortion) in cases:
    viewpoint_cam = types.SimpleNamespace(world_view_transform=rigid(seed), image_width=W, image_height=H, FoVx=fovx, FoVy=fovy)
    rendering = synthetic_rendering(W, H, g).requires_grad_(True)
    gt_image = torch.rand(3, H, W, generator=g)
    # ---- train.py:150-188, the same calls in the same order ----
    image = rendering[:3, :, :]
    Ll1 = l1_loss(image, gt_image)
    ssim_value = ssim(image, gt_image)
    rgb_loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim_value)
    distortion_map = rendering[8, :, :]
    distortion_loss = distortion_map.mean()
    depth = rendering[6, :, :]
    depth_normal, _ = depth_to_normal(viewpoint_cam, depth[None, ...])
    depth_normal = depth_normal.permute(2, 0, 1)
    render_normal = rendering[3:6, :, :]
    render_normal = torch.nn.functional.normalize(render_normal, p=2, dim=0)
    c2w = (viewpoint_cam.world_view_transform.T).inverse()
    normal2 = c2w[:3, :3] @ render_normal.reshape(3, -1)
    render_normal_world = normal2.reshape(3, *render_normal.shape[1:])
    normal_error = 1 - (render_normal_world * depth_normal).sum(dim=0)
    depth_normal_loss = normal_error.mean()
    loss = rgb_loss + depth_normal_loss * lambda_depth_normal + distortion_loss * lambda_distortion
    loss.backward()
    # -------------------------------------------------------------
    out[f"{tag}_cam"] = np.array([W, H, fovx, fovy], dtype=np.float64)
    out[f"{tag}_lambdas"] = np.array([lambda_dssim, lambda_depth_normal, lambda_distortion], dtype=np.float64)
    out[f"{tag}_wvt"] = viewpoint_cam.world_view_transform.numpy()
    out[f"{tag}_rendering"], out[f"{tag}_gt"] = rendering.detach().numpy(), gt_image.numpy()
    out[f"{tag}_terms"] = np.array([t.item() for t in (loss, Ll1, ssim_value, rgb_loss, depth_normal_loss, distortion_loss)], dtype=np.float64)
    out[f"{tag}_grad"] = rendering.grad.numpy()

path = os.path.join(HERE, "ref_train_loss_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
