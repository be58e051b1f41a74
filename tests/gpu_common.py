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
    """with forward_exact(): ... -- the forward blend's verification mode (every pair in the reference's own arithmetic, every output
    bit the oracle's) for the calls THIS THREAD makes inside the block: a per-call mode (GofRasterArgs.forward_exact, ABI 12;
    _backend.call_modes), nothing process-wide.  With `lib` (a second library instance: the audit / variant builds, or the host
    emulation, whose bindings build their own argument structs) the process-wide default of THAT library is switched and restored."""

    def __init__(self, on=True, lib=None):
        self.on, self.lib = on, lib

    def __enter__(self):
        if self.lib is None:
            from diff_gaussian_rasterization import _backend as B
            self.ctx = B.call_modes(forward_exact=bool(self.on))
            self.ctx.__enter__()
        else:
            self.prev = self.lib.gof_set_forward_exact(1 if self.on else 0)

    def __exit__(self, *exc):
        if self.lib is None:
            self.ctx.__exit__(*exc)
        else:
            self.lib.gof_set_forward_exact(self.prev)


# The forward blend's default mode (the reference's arithmetic without its two fp64 divisions per pair: csrc/gof_common.h,
# pair_nodiv_cc) against the verification mode.  t, alpha and T are the exact arithmetic's up to a last bit of one pair in ~1e7, so:
# every decision identical; colour / normal / depth / alpha channels and final_T bit-identical on all but a vanishing number of pixels
# (measured: none anywhere: profiles/r05_parity_report.md, 36 scenes incl. S1M / posed / clustered at both kernel sizes); the
# distortion channel, dist1 and dist2 carry the fp32 mapped depth's 1-2 ulp.  Measured on those 36 scenes (same report): channel 8
# within 2.5e-7 ABSOLUTE everywhere, i.e. within 6.3e-5 of the channel's maximum on every scene whose distortion reaches 2e-3 (S1M:
# 4.7e-5) -- north_star's "1e-4 relative" -- and up to 1.1e-3 of the maximum on the four scenes whose distortion is itself below 8e-4
# (lego10k, tiny, posed_tiny, sub_tile): the channel is a cancelling sum m^2 A + dist2 - 2 m dist1 of O(alpha) terms
# (forward.cu:553), so one fp32 ulp of a term (6e-8) is a fixed absolute, not a relative, quantity.  Asserted accordingly: channel 8
# within max(1e-4 x the channel's maximum, 3e-7 absolute); everything else within FAST_MODE_TOL.
FAST_MODE_TOL = 2e-6
FAST_MODE_CH8_REL, FAST_MODE_CH8_ABS = 1e-4, 3e-7


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
            if ch == 8:
                ch_max = float(np.abs(b[np.isfinite(b)]).max())
                assert d.max() <= max(FAST_MODE_CH8_REL * ch_max, FAST_MODE_CH8_ABS), ("distortion channel", float(d.max()), ch_max)
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
