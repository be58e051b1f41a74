"""Helpers shared by the GPU parity tests: run a synthetic scene through the product
(libgof_hip.so via the diff_gaussian_rasterization mirror) and through the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-opacity-fields_amd")
for p in (PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def to_dev(scene, device="cuda:0"):
    d = {}
    for k, v in scene.items():
        d[k] = torch.from_numpy(v).to(device) if isinstance(v, np.ndarray) else v
    return d


def settings_from(sd, debug=False, prefiltered=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=sd["H"], image_width=sd["W"], tanfovx=sd["tanfovx"], tanfovy=sd["tanfovy"],
        kernel_size=sd["kernel_size"], subpixel_offset=sd["subpixel_offset"], bg=sd["bg"],
        scale_modifier=sd["scale_modifier"], viewmatrix=sd["viewmatrix"], projmatrix=sd["projmatrix"],
        sh_degree=sd["sh_degree"], campos=sd["campos"], prefiltered=prefiltered, debug=debug)


def product_forward_raw(sd, **over):
    """Call the native forward directly; returns dict with outputs + workspaces."""
    from diff_gaussian_rasterization import _backend as B
    empty = torch.Tensor([])
    colors = over.get("colors_precomp", empty)
    sh = empty if "colors_precomp" in over else sd["shs"]
    cov = over.get("cov3D_precomp", empty)
    v2g = over.get("view2gaussian_precomp", empty)
    scales = over.get("scales", sd["scales"]); rot = over.get("rotations", sd["rotations"])
    args = (sd["bg"], sd["means3D"], colors, sd["opacities"], scales, rot, sd["scale_modifier"], cov, v2g,
            sd["viewmatrix"], sd["projmatrix"], sd["tanfovx"], sd["tanfovy"], sd["kernel_size"], sd["subpixel_offset"],
            sd["H"], sd["W"], sh, sd["sh_degree"], sd["campos"], over.get("prefiltered", False), over.get("debug", False))
    R, color, radii, geom, binning, img = B.rasterize_gaussians(*args, fused=over.get("fused", False))   # exact num_rendered for the parity checks
    view = B._View(*args)
    return dict(R=R, color=color, radii=radii, geom=geom, binning=binning, img=img, view=view, args=args)


def fetch(res, name):
    from diff_gaussian_rasterization import _backend as B
    return B.debug_fetch(name, res["view"], res["R"], res["geom"], res["binning"], res["img"]).cpu().numpy()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def rel_err(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)
