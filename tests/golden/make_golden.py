"""Generates tests/golden/ref_python_golden.npz by IMPORTING the reference's own Python
restatements from /root/reference (run in the build container only; the GPU box has no
/root/reference and only reads the committed .npz).

What is pinned (these are the only in-repo restatements of the hot path):
  * utils/sh_utils.py:57-112        eval_sh                  -> SH colour (K1)
  * scene/gaussian_model.py:202-260 get_view2gaussian        -> the 10 view2gaussian floats (K1)
  * scene/gaussian_model.py:65-69   covariance from scaling/rotation (utils/general_utils.py:62-110) -> cov3D (K1)
  * utils/graphics_utils.py:51-71   getProjectionMatrix       -> camera convention of tests/bench scenes
  * utils/tetmesh.py:141-190        marching_tetrahedra       -> marching-tets edges / faces (exact integers)

Missing third-party imports of scene/gaussian_model.py (plyfile, trimesh, simple_knn) are stubbed;
torch.zeros/ones(device='cuda') are redirected to the CPU.  No reference code is copied: the
functions are called where they lie.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "gaussian-opacity-fields_amd"))

# ---- stubs for absent third-party modules -------------------------------------------------------
for name in ("plyfile", "trimesh", "simple_knn", "simple_knn._C", "open3d", "cv2"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        sys.modules[name] = m
sys.modules["plyfile"].PlyData = object
sys.modules["plyfile"].PlyElement = object
sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: None

_zeros, _ones = torch.zeros, torch.ones


def _cpu(fn):
    def wrapped(*a, **k):
        k.pop("device", None)
        return fn(*a, **k)
    return wrapped


torch.zeros = _cpu(_zeros)
torch.ones = _cpu(_ones)

from utils.sh_utils import eval_sh                      # noqa: E402
from utils.general_utils import build_scaling_rotation, strip_symmetric  # noqa: E402
from utils.graphics_utils import getProjectionMatrix    # noqa: E402
from utils.tetmesh import marching_tetrahedra           # noqa: E402
from scene.gaussian_model import GaussianModel          # noqa: E402

import synthetic_scenes as S                            # noqa: E402

out = {}
sc = S.scene_lego_like(P=512, W=64, H=48, seed=11)
rng = np.random.default_rng(5)
# anisotropic scales and random rotations so that the rotation conventions are exercised
sc["scales"] = (sc["scales"] * np.exp(rng.normal(0, 0.4, sc["scales"].shape))).astype(np.float32)
q = rng.normal(0, 1, (512, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
sc["rotations"] = q.astype(np.float32)
for k in ("means3D", "scales", "rotations", "shs", "viewmatrix", "projmatrix", "campos", "opacities"):
    out["in_" + k] = sc[k]
out["in_tanfov"] = np.array([sc["tanfovx"], sc["tanfovy"]], dtype=np.float64)
out["in_WH"] = np.array([sc["W"], sc["H"]], dtype=np.int64)

means = torch.from_numpy(sc["means3D"]); scales = torch.from_numpy(sc["scales"]); rots = torch.from_numpy(sc["rotations"])
shs = torch.from_numpy(sc["shs"]); campos = torch.from_numpy(sc["campos"])

# eval_sh for every degree, as gaussian_renderer/__init__.py:83-92 calls it
dirs = means - campos[None]
dirs = dirs / dirs.norm(dim=1, keepdim=True)
shs_view = shs.transpose(1, 2).reshape(-1, 3, 16)
for deg in range(4):
    out["sh_rgb_deg%d" % deg] = torch.clamp_min(eval_sh(deg, shs_view, dirs) + 0.5, 0.0).numpy()

# covariance (scene/gaussian_model.py:65-69 with scaling_modifier)
for mod in (1.0, 0.7):
    L = build_scaling_rotation(mod * scales, rots)
    cov = L @ L.transpose(1, 2)
    out["cov3D_mod%.1f" % mod] = strip_symmetric(cov).numpy()


class _Mock:
    pass


m = _Mock()
m._rotation = rots
m.get_xyz = means
m.get_scaling_with_3D_filter = scales
out["view2gaussian"] = GaussianModel.get_view2gaussian(m, torch.from_numpy(sc["viewmatrix"])).numpy()

fovx = 2 * np.arctan(sc["tanfovx"]); fovy = 2 * np.arctan(sc["tanfovy"])
out["projection_T"] = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1).numpy()

# marching tetrahedra: Freudenthal grid, sdf = sphere with noise
verts, tets = S.freudenthal_tets(6, 5, 4)
sdf = (2.3 - np.linalg.norm(verts - np.array([3.0, 2.5, 2.0]), axis=1) + rng.normal(0, 0.2, verts.shape[0])).astype(np.float32)
vscale = rng.uniform(0.1, 1.0, verts.shape[0]).astype(np.float32)
res = marching_tetrahedra(torch.from_numpy(verts)[None], torch.from_numpy(tets), torch.from_numpy(sdf)[None], torch.from_numpy(vscale)[None, :, None])
(vpos, vsdf), vsc, faces, eids = [r[0] for r in res]
out["mt_verts"] = verts; out["mt_tets"] = tets; out["mt_sdf"] = sdf; out["mt_scales"] = vscale
out["mt_edge_pos"] = vpos.numpy(); out["mt_edge_sdf"] = vsdf.numpy(); out["mt_edge_scales"] = vsc.numpy()
out["mt_faces"] = faces.numpy(); out["mt_edge_ids"] = eids.numpy()
# the Kaolin docstring example (utils/tetmesh.py:164-180)
res = marching_tetrahedra(torch.tensor([[[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]]], dtype=torch.float), torch.tensor([[0, 1, 2, 3]]),
                          torch.tensor([[-1., -1., 0.5, 0.5]]), torch.ones(1, 4, 1))
out["mt_doc_faces"] = res[2][0].numpy(); out["mt_doc_edge_ids"] = res[3][0].numpy()

np.savez_compressed(os.path.join(HERE, "ref_python_golden.npz"), **out)
print("wrote", os.path.join(HERE, "ref_python_golden.npz"), {k: v.shape for k, v in out.items()})
