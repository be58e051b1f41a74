"""Generates tests/golden/ref_train_epilogue_golden.npz by EXECUTING the reference's own Python where it lies
(/root/reference; build container only -- the GPU box reads the committed .npz):

  * utils/loss_utils.py:17-63   l1_loss, ssim (value and autograd gradient w.r.t. img1; 3-D and batched 4-D inputs)
  * utils/depth_utils.py:6-35   depths_to_points, depth_to_normal (values and the autograd gradient of a random linear
                                functional w.r.t. depth).  The reference hard-codes `.cuda()` / `device='cuda'`; both are
                                redirected to the CPU for this run (Tensor.cuda -> identity, arange drops `device`).
  * torch.optim.Adam(lr=0.0, eps=1e-15) as scene/gaussian_model.py:360 builds it, with per-group lr, 3 steps
    (torch itself is the reference's dependency; single-tensor CPU implementation).
No reference code is copied.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

torch.Tensor.cuda = lambda self, *a, **k: self
_arange = torch.arange


def _arange_cpu(*a, **k):
    k.pop("device", None)
    return _arange(*a, **k)


torch.arange = _arange_cpu

from utils.loss_utils import l1_loss, ssim                       # noqa: E402
from utils.depth_utils import depths_to_points, depth_to_normal  # noqa: E402

out = {}
g = torch.Generator().manual_seed(1234)

# ---- ssim / l1 ----
for tag, shape in (("a", (3, 37, 53)), ("b", (3, 16, 16)), ("c", (1, 7, 9))):
    x = torch.rand(shape, generator=g, requires_grad=True)
    y = (x.detach() + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
    s = ssim(x, y)
    (gx,) = torch.autograd.grad(s, x)
    out[f"ssim_{tag}_x"], out[f"ssim_{tag}_y"] = x.detach().numpy(), y.numpy()
    out[f"ssim_{tag}_value"], out[f"ssim_{tag}_grad"] = s.detach().numpy(), gx.numpy()
    out[f"l1_{tag}_value"] = l1_loss(x.detach(), y).numpy()
xb = torch.rand((2, 3, 21, 19), generator=g, requires_grad=True)
yb = torch.rand((2, 3, 21, 19), generator=g)
sb = ssim(xb, yb, size_average=False)
wb = torch.tensor([0.7, -1.3])
(gb,) = torch.autograd.grad((sb * wb).sum(), xb)
out["ssim_batch_x"], out["ssim_batch_y"], out["ssim_batch_w"] = xb.detach().numpy(), yb.numpy(), wb.numpy()
out["ssim_batch_value"], out["ssim_batch_grad"] = sb.detach().numpy(), gb.numpy()


# ---- depth_to_normal ----
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


for tag, (W, H, fovx, fovy, seed) in (("a", (31, 23, 0.9, 0.7, 5)), ("b", (16, 16, 1.2, 1.2, 6)), ("c", (3, 3, 0.5, 0.6, 7))):
    view = types.SimpleNamespace(world_view_transform=rigid(seed), image_width=W, image_height=H, FoVx=fovx, FoVy=fovy)
    depth = (1.0 + 4.0 * torch.rand((1, H, W), generator=g)).requires_grad_(True)
    normals, points = depth_to_normal(view, depth)
    wn = torch.randn(normals.shape, generator=g)
    wp = torch.randn(points.shape, generator=g)
    (gd,) = torch.autograd.grad((normals * wn).sum() + (points * wp).sum(), depth)
    out[f"dn_{tag}_cam"] = np.array([W, H, fovx, fovy], dtype=np.float64)
    out[f"dn_{tag}_wvt"] = view.world_view_transform.numpy()
    out[f"dn_{tag}_depth"] = depth.detach().numpy()
    out[f"dn_{tag}_normals"], out[f"dn_{tag}_points"] = normals.detach().numpy(), points.detach().numpy()
    out[f"dn_{tag}_wn"], out[f"dn_{tag}_wp"], out[f"dn_{tag}_grad"] = wn.numpy(), wp.numpy(), gd.numpy()
    out[f"dn_{tag}_points_flat"] = depths_to_points(view, depth.detach()).numpy()

# ---- Adam as gaussian_model.py:360 configures it ----
ps = [torch.randn(n, generator=g).requires_grad_(True) for n in (1000, 4099, 17)]
lrs = [1.6e-4, 2.5e-3, 5e-2]
opt = torch.optim.Adam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))], lr=0.0, eps=1e-15,
                       foreach=False)
out["adam_p0"] = np.concatenate([p.detach().numpy() for p in ps])
out["adam_lrs"] = np.array(lrs)
out["adam_sizes"] = np.array([p.numel() for p in ps])
for step in range(3):
    grads = [torch.randn(p.shape, generator=g) * (10.0 ** (step - 2)) for p in ps]
    for p, gr in zip(ps, grads):
        p.grad = gr.clone()
    opt.step()
    out[f"adam_g{step}"] = np.concatenate([gr.numpy() for gr in grads])
    out[f"adam_p{step + 1}"] = np.concatenate([p.detach().numpy() for p in ps])
out["adam_m3"] = np.concatenate([opt.state[p]["exp_avg"].numpy() for p in ps])
out["adam_v3"] = np.concatenate([opt.state[p]["exp_avg_sq"].numpy() for p in ps])

# ---- GaussianModel.compute_3D_filter (scene/gaussian_model.py:262-311), the reference's own method, run on CPU tensors ----
for name in ("plyfile", "trimesh", "simple_knn", "simple_knn._C", "open3d", "cv2"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["plyfile"].PlyData = object
sys.modules["plyfile"].PlyElement = object
sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: None
from scene.gaussian_model import GaussianModel                   # noqa: E402

rng = np.random.default_rng(77)
P3 = 4000
xyz3 = rng.uniform(-1.5, 1.5, (P3, 3)).astype(np.float32)
xyz3[-60:] *= 300.0                                # far-away centres: most of them are seen by no camera (gaussian_model.py:306)
xyz3 = torch.from_numpy(xyz3)
cams3 = []
for i in range(7):
    M = rigid(100 + i).T.numpy()                   # world-to-camera [R|t]
    Rw2c, t = M[:3, :3].astype(np.float64), M[:3, 3].astype(np.float64)
    t = t * 0.3 + np.array([0, 0, 3.0])            # keep most of the cloud in front of the camera
    W3, H3 = (640, 480) if i % 2 else (800, 528)
    cams3.append(types.SimpleNamespace(R=Rw2c.T.copy(), T=t.copy(), focal_x=500.0 + 40 * i, focal_y=480.0 + 35 * i, image_width=W3, image_height=H3))
fake = types.SimpleNamespace(get_xyz=xyz3)
GaussianModel.compute_3D_filter(fake, cams3)
out["f3d_xyz"] = xyz3.numpy()
out["f3d_cams"] = np.stack([np.concatenate([c.R.reshape(9), c.T.reshape(3), [c.focal_x, c.focal_y, c.image_width, c.image_height]]) for c in cams3])
out["f3d_filter"] = fake.filter_3D.numpy()

# ---- the derived tensors render() reads: GaussianModel.get_scaling_with_3D_filter / get_opacity_with_3D_filter / get_rotation,
#      the reference's own property bodies on a stand-in object (scene/gaussian_model.py:152-194) ----
Pa = 3000
Stand = type("Stand", (), {k: getattr(GaussianModel, k) for k in ("get_scaling", "get_scaling_with_3D_filter", "get_opacity_with_3D_filter",
                                                                  "get_rotation", "setup_functions")})   # the reference's own property objects
act = Stand()
act.setup_functions()
act._scaling = (torch.randn((Pa, 3), generator=g) * 1.5 - 3.0).requires_grad_(True)
act._opacity = (torch.randn((Pa, 1), generator=g) * 2.0).requires_grad_(True)
act._rotation = torch.randn((Pa, 4), generator=g).requires_grad_(True)
act.filter_3D = torch.rand((Pa, 1), generator=g) * 0.05
sc_f = act.get_scaling_with_3D_filter
op_f = act.get_opacity_with_3D_filter
ro_n = act.get_rotation
w_s, w_o, w_r = torch.randn(sc_f.shape, generator=g), torch.randn(op_f.shape, generator=g), torch.randn(ro_n.shape, generator=g)
gs, go_, gr_ = torch.autograd.grad((sc_f * w_s).sum() + (op_f * w_o).sum() + (ro_n * w_r).sum(), [act._scaling, act._opacity, act._rotation])
for k, v in (("raw_scaling", act._scaling), ("raw_opacity", act._opacity), ("raw_rotation", act._rotation), ("filter_3D", act.filter_3D),
             ("scaling", sc_f), ("opacity", op_f), ("rotation", ro_n), ("w_s", w_s), ("w_o", w_o), ("w_r", w_r),
             ("g_scaling", gs), ("g_opacity", go_), ("g_rotation", gr_)):
    out["act_" + k] = v.detach().numpy()

path = os.path.join(HERE, "ref_train_epilogue_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
