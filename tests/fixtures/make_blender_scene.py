"""Generator of a tiny NeRF-synthetic ("Blender") format scene for the end-to-end runs of the reference's UNCHANGED scripts
(tests/test_e2e_scripts_gpu.py): `transforms_train.json` / `transforms_test.json` + PNG images + `points3d.ply`, exactly what
scene/dataset_readers.py:184-260 reads.  There is no dataset offline, so the images are renderings of a known set of ground-truth
Gaussians (a few coloured blobs around the origin) by the product rasterizer itself -- a scene the optimiser can actually fit,
which makes "the loss goes down and the PSNR goes up" a meaningful end-to-end statement.

    python tests/fixtures/make_blender_scene.py <out_dir> [--views 24] [--size 160 120]      (needs the GPU)
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "e2e_shims")):
    if p not in sys.path:
        sys.path.append(p)

FOVX = 0.6911112070083618            # camera_angle_x of the NeRF-synthetic scenes


def ground_truth(seed=0, n=3000, scale=0.035):
    """A few anisotropic blobs of small Gaussians inside the [-1.3, 1.3]^3 box the reader's random initialisation assumes."""
    rng = np.random.default_rng(seed)
    centres = np.array([[0.0, 0.0, 0.0], [0.7, 0.2, -0.3], [-0.6, -0.4, 0.4], [0.1, 0.6, 0.6]])
    radii = np.array([[0.55, 0.4, 0.45], [0.25, 0.3, 0.2], [0.3, 0.2, 0.3], [0.2, 0.2, 0.35]])
    base = np.array([[0.9, 0.3, 0.2], [0.2, 0.8, 0.3], [0.2, 0.4, 0.9], [0.9, 0.8, 0.2]])
    which = rng.integers(0, len(centres), n)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    shell = rng.uniform(0.85, 1.0, (n, 1))                       # points near the surface of each ellipsoid
    means = centres[which] + d * shell * radii[which]
    colors = np.clip(base[which] + 0.25 * d * np.array([1.0, -1.0, 0.5]) + rng.normal(0, 0.03, (n, 3)), 0.02, 0.98)
    scales = np.exp(rng.normal(math.log(scale), 0.25, (n, 3)))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return dict(means3D=means.astype(np.float32), colors=colors.astype(np.float32), scales=scales.astype(np.float32),
                rotations=q.astype(np.float32), opacities=np.full((n, 1), 0.95, np.float32))


def ground_truth_large(seed=0, n=250_000, scale=0.009):
    """The structured scene of the config-3-shaped evidence run (tests/devtools/dev_r6_trajectory.py): a dozen ellipsoid shells of
    different sizes, a tilted ground disc and three thin rods, every surface carrying a high-frequency colour pattern (stripes and
    checks of 0.04-0.1 units: sub-splat detail at 800x800 that the optimiser can only fit by densifying), drawn with `n` small
    anisotropic Gaussians.  Same dictionary as ground_truth()."""
    rng = np.random.default_rng(seed)
    ne = 12
    centres = rng.uniform(-0.85, 0.85, (ne, 3)) * np.array([1.0, 0.6, 1.0])
    centres[0] = 0.0
    radii = rng.uniform(0.12, 0.34, (ne, 3))
    radii[0] = [0.5, 0.38, 0.42]
    base = rng.uniform(0.15, 0.95, (ne, 3))
    share = np.concatenate([np.prod(radii, 1) ** (2.0 / 3.0), [0.9, 0.12, 0.12, 0.12]])      # ~ surface area; disc; rods
    which = rng.choice(ne + 4, n, p=share / share.sum())
    means = np.empty((n, 3)); colors = np.empty((n, 3)); nrm = np.empty((n, 3))
    for k in range(ne):
        m = which == k
        d = rng.normal(size=(m.sum(), 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        means[m] = centres[k] + d * radii[k] * rng.uniform(0.97, 1.0, (m.sum(), 1))
        nrm[m] = d
        stripes = 0.5 + 0.5 * np.sign(np.sin((25.0 + 6.0 * k) * d[:, 1]) * np.sin((18.0 + 5.0 * k) * np.arctan2(d[:, 0], d[:, 2])))
        colors[m] = base[k] * (0.35 + 0.65 * stripes[:, None]) + 0.15 * d
    m = which == ne                                   # ground disc y = 0.62 (y is down in the camera convention of look_at_pose), radius 1.25
    r = 1.25 * np.sqrt(rng.uniform(0, 1, m.sum())); a = rng.uniform(0, 2 * math.pi, m.sum())
    means[m] = np.stack([r * np.cos(a), 0.62 + 0.08 * np.sin(3.0 * r * np.cos(a)), r * np.sin(a)], 1)
    nrm[m] = [0.0, -1.0, 0.0]
    check = (np.floor(means[m][:, 0] / 0.09) + np.floor(means[m][:, 2] / 0.09)) % 2
    colors[m] = np.where(check[:, None] > 0, [0.85, 0.85, 0.8], [0.15, 0.2, 0.3])
    for j in range(3):                                # rods
        m = which == ne + 1 + j
        t = rng.uniform(-1.0, 1.0, m.sum()); a = rng.uniform(0, 2 * math.pi, m.sum())
        axis = np.eye(3)[j]; u = np.eye(3)[(j + 1) % 3]; v = np.eye(3)[(j + 2) % 3]
        off = np.array([0.45, -0.35, -0.5])[j] * u + np.array([-0.4, 0.5, 0.3])[j] * v
        means[m] = t[:, None] * axis + off + 0.035 * (np.cos(a)[:, None] * u + np.sin(a)[:, None] * v)
        nrm[m] = np.cos(a)[:, None] * u + np.sin(a)[:, None] * v
        colors[m] = np.where((np.floor(t / 0.06) % 2)[:, None] > 0, np.eye(3)[j] * 0.9 + 0.05, [0.9, 0.9, 0.9])
    colors = np.clip(colors + rng.normal(0, 0.02, (n, 3)), 0.02, 0.98)
    # flat splats lying in the surface: two tangent axes of `scale`, the normal axis a fifth of it; rotation = [t1, t2, normal]
    t1 = np.cross(nrm, rng.normal(size=(n, 3))); t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    Rm = np.stack([t1, t2, nrm], 2)
    q = np.empty((n, 4))
    tr = Rm[:, 0, 0] + Rm[:, 1, 1] + Rm[:, 2, 2]
    q[:, 0] = 0.5 * np.sqrt(np.maximum(1e-12, 1.0 + tr))
    q[:, 1] = (Rm[:, 2, 1] - Rm[:, 1, 2]) / (4.0 * q[:, 0]); q[:, 2] = (Rm[:, 0, 2] - Rm[:, 2, 0]) / (4.0 * q[:, 0]); q[:, 3] = (Rm[:, 1, 0] - Rm[:, 0, 1]) / (4.0 * q[:, 0])
    bad = tr < -0.9                                   # (half-turns: the formula above loses its digits; any unit quaternion draws a valid splat)
    q[bad] = rng.normal(size=(bad.sum(), 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    scales = np.exp(rng.normal(math.log(scale), 0.25, (n, 3))) * np.array([1.0, 1.0, 0.2])
    return dict(means3D=means.astype(np.float32), colors=colors.astype(np.float32), scales=scales.astype(np.float32),
                rotations=q.astype(np.float32), opacities=np.full((n, 1), 0.95, np.float32))


def look_at_pose(theta, phi, radius=4.0):
    """(R_c2w in the reference's camera convention: x right, y down, z forward; camera centre)."""
    c = radius * np.array([math.cos(phi) * math.sin(theta), -math.sin(phi), math.cos(phi) * math.cos(theta)])
    fwd = -c / np.linalg.norm(c)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    return np.stack([right, down, fwd], 1), c


def render_view(gt, R_c2w, centre, W, H, bg, device="cuda:0"):
    import torch
    import synthetic_scenes as S
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    focal = W / (2 * math.tan(FOVX / 2))
    fovy = 2 * math.atan(H / (2 * focal))
    cam = S.camera(W, H, FOVX, fovy, R=R_c2w, T=-R_c2w.T @ centre)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)    # noqa: E731
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], kernel_size=0.0,
        subpixel_offset=torch.zeros((H, W, 2), device=device), bg=t(np.asarray(bg, np.float32)), scale_modifier=1.0,
        viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=0, campos=t(cam["campos"]), prefiltered=False, debug=False)
    with torch.no_grad():
        means = t(gt["means3D"])
        color, _ = GaussianRasterizer(settings)(means3D=means, means2D=torch.zeros_like(means), colors_precomp=t(gt["colors"]),
                                                opacities=t(gt["opacities"]), scales=t(gt["scales"]), rotations=t(gt["rotations"]))
    return color[:3].clamp(0, 1).permute(1, 2, 0).cpu().numpy()


def make_scene(out_dir, n_train=24, n_test=4, W=160, H=120, seed=0, n_init=6000, white_background=False, n_gt=3000, gt_scale=0.035, large=False):
    from PIL import Image
    from plyfile import PlyData, PlyElement
    os.makedirs(os.path.join(out_dir, "train"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "test"), exist_ok=True)
    gt = ground_truth_large(seed, n_gt, gt_scale) if large else ground_truth(seed, n_gt, gt_scale)
    rng = np.random.default_rng(seed + 1)
    bg = (1.0, 1.0, 1.0) if white_background else (0.0, 0.0, 0.0)
    for split, n in (("train", n_train), ("test", n_test)):
        frames = []
        for i in range(n):
            theta = 2 * math.pi * (i + (0.5 if split == "test" else 0.0)) / n
            phi = 0.25 + 0.35 * math.sin(3.1 * i + (1.0 if split == "test" else 0.0)) + rng.uniform(-0.05, 0.05)
            R, c = look_at_pose(theta, phi)
            img = render_view(gt, R, c, W, H, bg)
            Image.fromarray((img * 255.0 + 0.5).astype(np.uint8), "RGB").save(os.path.join(out_dir, split, "r_%d.png" % i))
            c2w = np.eye(4)
            c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = R[:, 0], -R[:, 1], -R[:, 2], c     # OpenGL axes: y up, z back (dataset_readers.py:196-198 flips them back)
            frames.append({"file_path": "./%s/r_%d" % (split, i), "transform_matrix": c2w.tolist()})
        with open(os.path.join(out_dir, "transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": FOVX, "frames": frames}, f)
    # initial point cloud (dataset_readers.py:241-256 would draw 100k random points; a smaller cloud keeps the test short)
    xyz = rng.uniform(-1.3, 1.3, (n_init, 3)).astype(np.float32)
    el = np.empty(n_init, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    el["x"], el["y"], el["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    el["nx"] = el["ny"] = el["nz"] = 0
    rgb = rng.integers(100, 156, (n_init, 3))
    el["red"], el["green"], el["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    PlyData([PlyElement.describe(el, "vertex")]).write(os.path.join(out_dir, "points3d.ply"))
    return out_dir


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--size", type=int, nargs=2, default=[160, 120])
    ap.add_argument("--gt", type=int, default=3000, help="ground-truth Gaussians the images are rendered from")
    ap.add_argument("--gt-scale", type=float, default=0.035)
    ap.add_argument("--init", type=int, default=6000, help="points of the initial cloud (points3d.ply)")
    ap.add_argument("--test-views", type=int, default=4)
    ap.add_argument("--large", action="store_true", help="the structured ground truth of the config-3-shaped evidence run (ground_truth_large)")
    a = ap.parse_args()
    make_scene(a.out, n_train=a.views, n_test=a.test_views, W=a.size[0], H=a.size[1], n_gt=a.gt, gt_scale=a.gt_scale, n_init=a.init, large=a.large)
    print("wrote", a.out)
