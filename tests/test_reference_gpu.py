"""GPU tests against the REAL reference: the CUDA sources of /root/reference compiled for gfx950 by
oracle/build_ref.sh (oracle/_ref/libgof_cudaref*.so, built in the container, shipped with the snapshot).

  * nofma build  vs ORACLE : pins the CPU restatement to the reference's own source, stage by stage --
    K1 outputs, sort keys and the sorted list BIT-EXACT (same IEEE operations, no contraction); the
    blended image differs only through exp() (device expf vs the oracle's exp), bounded at 2e-5.
  * default build vs PRODUCT: what a user switching from the reference sees on the same GPU: same
    visible set, identical tile lists up to depth-key ties, image / gradients within the north_star
    tolerance (1e-4 relative) on the well-conditioned scene.
"""
import numpy as np
import pytest
import torch

import oracle_binding as ob
import reference_binding as rb
import synthetic_scenes as S
from gpu_common import bits, fetch, product_forward_raw, settings_from, to_dev

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not rb.available(), reason="oracle/_ref not built (needs /root/reference at build time)")]


def scene():
    return S.scene_frustum(20_000, W=320, H=208, focal=240.0, seed=0, kernel_size=0.1, bg=(0.1, 0.2, 0.3))


def test_oracle_is_pinned_to_reference_source_stage_by_stage():
    sc = scene()
    o = ob.OracleScene(sc)
    oc, orad = o.forward()
    ref = rb.Reference(to_dev(sc), "_nofma")
    rc, rrad = ref.forward()
    assert ref.R == o.num_rendered()
    assert np.array_equal(rrad, orad)
    vis = orad > 0
    P = len(orad)
    for name in ("depths", "means2D", "cov3D", "conic_opacity", "rgb", "view2gaussian", "clamped"):
        a = ref.fetch(name).reshape(P, -1)[vis]; b = o.fetch(name).reshape(P, -1)[vis]
        assert np.array_equal(bits(a), bits(b)), (name, int((bits(a) != bits(b)).sum()), a.size)
    for name in ("tiles_touched", "point_offsets", "point_list_keys", "point_list", "ranges"):
        assert np.array_equal(ref.fetch(name), o.fetch(name)), name
    # blend: identical except for exp(); contributor counts agree for (almost) every pixel
    assert np.abs(rc - oc).max() <= 2e-5 * max(1.0, np.abs(oc).max())
    nc_r = ref.fetch("n_contrib"); nc_o = o.fetch("n_contrib")
    assert (nc_r != nc_o).mean() < 1e-3


def test_oracle_backward_matches_reference_backward():
    sc = scene()
    o = ob.OracleScene(sc)
    oc, orad = o.forward()
    ref = rb.Reference(to_dev(sc), "_nofma")
    ref.forward()
    dL = np.random.default_rng(1).normal(size=oc.shape).astype(np.float32)
    gr = ref.backward(dL)
    go = o.backward(dL)
    for k in ("means2D", "colors", "opacity", "view2gaussian"):
        tol = 1e-4 * np.abs(go[k]).max()
        assert np.abs(gr[k].reshape(go[k].shape) - go[k]).max() <= tol, k
    assert not gr["cov3D"].any()
    # per-Gaussian stage on the REFERENCE's own dL_dview2gaussian / dL_dcolors: the restatement must reproduce it
    iso = o.preprocess_backward(gr["view2gaussian"], gr["colors"])
    for k in ("means3D", "sh", "scales", "rotations"):
        ref_v = gr[k].reshape(iso[k].shape)
        assert np.abs(ref_v - iso[k]).max() <= 1e-5 * max(np.abs(iso[k]).max(), 1e-20), k


def _err(a, b):
    scale = np.abs(b).reshape(9, -1).max(axis=1)[:, None, None] + 1e-12
    return np.abs(a - b) / scale


def test_product_vs_reference_default_build_within_the_references_own_fma_band():
    """The ray-Gaussian evaluation is ill-conditioned in fp32: the reference's OWN output
    moves by up to 1e-1 relative at single pixels when the compiler contracts multiply-adds differently.
    So the product (no contraction, = the nofma build bit for bit up to exp) is held to the band spanned by
    the reference's two builds, and to the north_star tolerance (1e-4) at the median."""
    sc = scene()
    sd = to_dev(sc)
    ref = rb.Reference(sd, "")
    rc, rrad = ref.forward()
    ref2 = rb.Reference(sd, "_nofma")
    rc2, rrad2 = ref2.forward()
    res = product_forward_raw(sd)
    torch.cuda.synchronize()
    prad = res["radii"].cpu().numpy()
    # against the nofma build the integer stage is exact
    assert np.array_equal(prad, rrad2) and res["R"] == ref2.R
    assert np.array_equal(fetch(res, "point_list").view(np.uint32), ref2.fetch("point_list"))
    # against the default build: radii may flip by one at exact ceil() boundaries -> count, do not fail
    assert (prad != rrad).mean() < 1e-3 and ((prad > 0) == (rrad > 0)).mean() > 0.9999
    assert abs(res["R"] - ref.R) <= 1e-3 * ref.R
    pc = res["color"].cpu().numpy()
    band = _err(rc2, rc)            # reference(nofma) vs reference(default): the reference's own compiler band
    mine = _err(pc, rc)             # product vs reference(default)
    qs = [50, 99, 99.9]
    pb, pm = np.percentile(band, qs), np.percentile(mine, qs)
    assert pm[0] < 1e-4, pm
    assert (pm <= 2.0 * pb + 1e-5).all(), (pm, pb)
    assert np.abs(pc - rc2).max() <= 2e-5 * max(1.0, np.abs(rc2).max())       # vs the nofma build: only exp() (and the normals' rsq) differ
    # gradients against the default build
    dL = np.random.default_rng(2).normal(size=rc.shape).astype(np.float32)
    gr = ref.backward(dL)
    gr2 = ref2.backward(dL)
    from test_parity_gpu import _product_backward
    gp = _product_backward(res, dL)
    for k in ("means2D", "colors", "opacity", "view2gaussian", "sh"):
        b = gr[k]
        mine_l2 = np.linalg.norm(gp[k].reshape(b.shape) - b) / (np.linalg.norm(b) + 1e-30)
        band_l2 = np.linalg.norm(gr2[k] - b) / (np.linalg.norm(b) + 1e-30)
        assert mine_l2 <= 2.0 * band_l2 + 1e-5, (k, mine_l2, band_l2)
        nofma_l2 = np.linalg.norm(gp[k].reshape(b.shape) - gr2[k]) / (np.linalg.norm(gr2[k]) + 1e-30)
        assert nofma_l2 < 1e-4, (k, nofma_l2)


def test_integrate_matches_reference():
    sc = S.scene_frustum(3000, W=96, H=64, focal=70.0, seed=8, kernel_size=0.1)
    pts = S.tetra_points(sc)[::3]
    sd = to_dev(sc)
    ref = rb.Reference(sd, "_nofma")
    rc, ral, rcol, rrad = ref.integrate(pts)
    o = ob.OracleScene(sc)
    oc, oal, ocol, orad = o.integrate(pts)
    assert np.array_equal(rrad, orad)
    assert np.array_equal(rc[8], oc[8])                       # points per pixel
    di = np.abs(rc - oc)   # an alpha >= 1/255 decision flipped by the last ulp of exp() moves a pixel by <= alpha*T*c ~ 1e-2
    assert np.percentile(di, 99.9) < 5e-5 and di.max() < 2e-2, (np.percentile(di, [99, 99.9]), di.max())
    # alpha: identical algorithm, exp() differs in the last ulp; a threshold flip moves alpha by <= 1/255
    d = np.abs(ral - oal)
    assert np.percentile(d, 99.9) < 1e-5 and d.max() < 5e-3, (np.percentile(d, [50, 99, 99.9]), d.max())
    dc = np.abs(rcol - ocol)
    assert np.percentile(dc, 99.9) < 5e-5 and dc.max() < 2e-2
