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


class forward_exact:
    """with forward_exact(): ... -- the forward blend's verification mode (gof_set_forward_exact: every pair in the reference's own
    arithmetic, every output bit the oracle's) for the calls inside; the previous mode is restored."""

    def __init__(self, on=True, lib=None):
        self.on, self.lib = on, lib

    def __enter__(self):
        if self.lib is None:
            from diff_gaussian_rasterization import _backend as B
            self.lib = B.lib
        self.prev = self.lib.gof_set_forward_exact(1 if self.on else 0)

    def __exit__(self, *exc):
        self.lib.gof_set_forward_exact(self.prev)


# The forward blend's default mode (the reference's arithmetic without its two fp64 divisions per pair: csrc/gof_common.h,
# pair_nodiv_cc) against the verification mode.  t, alpha and T are the exact arithmetic's up to a last bit of one pair in ~1e7, so:
# every decision identical; colour / normal / depth / alpha channels and final_T bit-identical on all but a vanishing number of pixels
# (measured: none anywhere, profiles/r04_forward_modes.md); the distortion channel, dist1 and dist2 carry the fp32 mapped depth's
# 1-2 ulp: measured <= 2.5e-7 absolute, asserted at 2e-6 (mapped depth <= 1).
FAST_MODE_TOL = 2e-6


def assert_fast_mode_matches_exact(fast, exact):
    """`fast`, `exact`: dicts of numpy arrays from the two forward modes on the same inputs -- color [9,H,W], final_T [4*HW],
    n_contrib, contrib_hash, tile_cost, radii.  Every DECISION must be the exact arithmetic's (integer arrays equal bit for bit, the
    contributor masks by a position-sensitive checksum per tile); channels 0-7 and T bit-identical on all but <= max(2, 1e-6 HW)
    pixels and within FAST_MODE_TOL there; the distortion quantities within FAST_MODE_TOL."""
    for k in ("radii", "n_contrib", "contrib_hash", "tile_cost"):
        assert np.array_equal(fast[k], exact[k]), (k, int((fast[k] != exact[k]).sum()))
    cf, cx = np.asarray(fast["color"], np.float32), np.asarray(exact["color"], np.float32)
    HW = cf.shape[1] * cf.shape[2]
    allowed = max(2, int(1e-6 * HW))
    for ch in range(9):
        a, b = cf[ch].ravel(), cx[ch].ravel()
        differ = bits(a) != bits(b)
        if ch != 8:
            assert int(differ.sum()) <= allowed, ("channel", ch, "pixels with different bits", int(differ.sum()))
        if differ.any():
            assert np.isfinite(a[differ]).all() and np.isfinite(b[differ]).all(), ("channel", ch, "a non-finite value differs")
            d = np.abs(a[differ].astype(np.float64) - b[differ])
            scale = 1.0 if ch == 8 else max(1.0, float(np.abs(b[np.isfinite(b)]).max()))
            assert d.max() <= FAST_MODE_TOL * scale, ("channel", ch, float(d.max()), scale)
    tf, tx = np.asarray(fast["final_T"], np.float32), np.asarray(exact["final_T"], np.float32)
    assert int((bits(tf[:HW]) != bits(tx[:HW])).sum()) <= allowed and np.abs(tf[:HW].astype(np.float64) - tx[:HW]).max() <= FAST_MODE_TOL
    for q in (1, 2, 3):                                                                          # dist1, dist2, distortion before normalisation
        m = max(1.0, float(np.abs(tx[q * HW:(q + 1) * HW]).max()))
        assert np.abs(tf[q * HW:(q + 1) * HW].astype(np.float64) - tx[q * HW:(q + 1) * HW]).max() <= FAST_MODE_TOL * m, q


def forward_mode_arrays(res):
    """the arrays assert_fast_mode_matches_exact compares, from a product_forward_raw result"""
    return dict(color=res["color"].cpu().numpy(), radii=res["radii"].cpu().numpy(), final_T=fetch(res, "final_T"),
                n_contrib=fetch(res, "n_contrib"), contrib_hash=fetch(res, "contrib_hash"), tile_cost=fetch(res, "tile_cost"))


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
