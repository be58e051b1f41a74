"""GPU tests against the REAL reference: the CUDA sources of /root/reference compiled for gfx950 by
oracle/build_ref.sh (oracle/_ref/libgof_cudaref*.so, built in the container, shipped with the snapshot).

  * nofma build  vs ORACLE : pins the CPU restatement to the reference's own source, stage by stage --
    K1 outputs, sort keys and the sorted list BIT-EXACT (same IEEE operations, no contraction); the
    blended image differs only through exp() (device expf vs the oracle's exp), bounded at 2e-5.
  * default build vs PRODUCT: what a user switching from the reference sees on the same GPU: same
    visible set, identical tile lists up to depth-key ties, image / gradients within the north_star
    tolerance (1e-4 relative) on the well-conditioned scene.
"""
import json
import os

import numpy as np
import pytest
import torch

import oracle_binding as ob
import reference_binding as rb
import synthetic_scenes as S
from gpu_common import bits, fetch, product_forward_raw, settings_from, to_dev

pytestmark = pytest.mark.gpu


def test_the_compiled_reference_travelled_to_the_gpu_box():
    """No silent skip: a GPU run without oracle/_ref (built by __graft_entry__.build() / oracle/build_ref.sh where /root/reference
    exists, git-ignored, shipped with the snapshot) would lose the pin of the oracle to the reference's own source."""
    for variant in ("", "_nofma"):
        assert rb.available(variant), "oracle/_ref/libgof_cudaref%s.so is missing: run oracle/build_ref.sh where /root/reference exists" % variant


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


BAND_SCENES = ["base20k", "posed_ragged", "posed_mod2", "posed_mod05_ks01", "clustered150k", "posed_clustered150k", "posed_mid100k"]
BAND_REPORT = {}          # scene -> measured figures (written to gpurun_out/reference_default_build_band.json by the last parametrisation)


@pytest.mark.parametrize("name", BAND_SCENES)
def test_product_vs_reference_default_build_within_the_references_own_fma_band(name):
    """What a user who switches backends sees: the product against the reference's DEFAULT (FMA-contracting) build.
    The ray-Gaussian evaluation is ill-conditioned in fp32: the reference's OWN output
    moves by up to 1e-1 relative at single pixels when the compiler contracts multiply-adds differently.
    So the product (no contraction, = the nofma build bit for bit up to exp) is held to the band spanned by
    the reference's two builds, and to the north_star tolerance (1e-4) at the median -- on posed cameras, scale_modifier != 1 and
    heavy-tailed scenes as well (round 4; round 3 asserted it on one 20 k-Gaussian scene)."""
    from test_parity_gpu import SCENES
    sc = scene() if name == "base20k" else SCENES[name]()
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
        BAND_REPORT.setdefault(name, {}).setdefault("gradients_rel_l2", {})[k] = {"product_vs_default": float(mine_l2), "reference_band": float(band_l2), "product_vs_nofma": float(nofma_l2)}
    BAND_REPORT[name].update(image_percentiles_50_99_99p9={"product_vs_default": [float(x) for x in pm], "reference_band": [float(x) for x in pb]},
                             worst_ratio_product_over_band=float(np.max(pm / np.maximum(pb, 1e-12))),
                             radii_differing_fraction=float((prad != rrad).mean()), R_product=int(res["R"]), R_reference_default=int(ref.R))
    if name == BAND_SCENES[-1]:
        import json, os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(BAND_REPORT, open(os.path.join(out, "reference_default_build_band.json"), "w"), indent=1)


def test_s1m_forward_vs_reference_default_build():
    """BASELINE's full-size configuration, forward: the product against the reference's default (FMA-contracting) build -- radii
    and instance count up to ceil() flips, the image inside the band the reference's two builds span at the median, the 99th and the
    99.9th percentile (at this scene's conditioning -- distance / scale ~ 480 -- the reference's own two builds differ by 2e-4 at the
    MEDIAN pixel and 3e-2 at the 99.9th percentile: the 1e-4 median bar of the smaller scenes does not exist here)."""
    sc = S.scene_frustum(1_000_000, seed=0)
    sd = to_dev(sc)
    ref = rb.Reference(sd, "")
    rc, rrad = ref.forward()
    R_default = ref.R
    ref2 = rb.Reference(sd, "_nofma")
    rc2, rrad2 = ref2.forward()
    del ref, ref2
    res = product_forward_raw(sd)
    torch.cuda.synchronize()
    prad = res["radii"].cpu().numpy()
    assert np.array_equal(prad, rrad2)
    assert (prad != rrad).mean() < 1e-3 and abs(res["R"] - R_default) <= 1e-3 * R_default
    pc = res["color"].cpu().numpy()
    band, mine = _err(rc2, rc), _err(pc, rc)
    qs = [50, 99, 99.9]
    pb, pm = np.percentile(band, qs), np.percentile(mine, qs)
    assert (pm <= 1.05 * pb + 1e-6).all(), (pm, pb)                            # measured: equal to the band to four digits
    assert np.abs(pc - rc2).max() <= 2e-5 * max(1.0, np.abs(rc2).max())


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


# ---- the pin, widened (round 2): the whole scene table of test_parity_gpu.py -- kernel_size 0.0 (the reference's default and what
# ---- every published run uses) and 0.1, SH degrees 0-3, > 256-entry lists, ragged sizes, the cull stress scene -- and S1M ----

def _pin_forward(sc, image_tol=2e-5):
    """oracle vs the reference's own source (no-contraction build): every integer stage and every K1 float bit-exact, the blended
    image equal up to exp() (device expf vs the oracle's exp), contributor counts equal on > 99.9 % of the pixels."""
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
    d = np.abs(rc - oc)
    scale = max(1.0, np.abs(oc).max())
    # a pixel where the last ulp of exp() flips an alpha >= 1/255 or T < 1e-4 decision moves by up to alpha*T*c: count those apart
    assert np.percentile(d, 99.9) <= image_tol * scale, np.percentile(d, [50, 99, 99.9, 100])
    nc_r = ref.fetch("n_contrib"); nc_o = o.fetch("n_contrib")
    assert (nc_r != nc_o).mean() < 2e-3
    assert (d > 100 * image_tol * scale).mean() <= 4.0 * max((nc_r != nc_o).mean(), 1e-6), ((d > 100 * image_tol * scale).mean(), (nc_r != nc_o).mean())
    return o, ref


@pytest.mark.parametrize("name", ["tiny", "one", "small_ks0", "small_ks01", "lego10k", "ragged", "long_lists", "mid100k", "stress_box",
                                  "posed_tiny", "posed_small_ks01", "posed_ragged", "posed_long_lists", "posed_mid100k", "posed_stress_box",
                                  "posed_mod2", "posed_mod05_ks01", "clustered150k", "posed_clustered150k"])
def test_oracle_pinned_to_reference_on_the_scene_table(name):
    from test_parity_gpu import SCENES
    _pin_forward(SCENES[name]())


@pytest.mark.parametrize("name", ["posed_small_ks01", "posed_ragged", "posed_mod2", "posed_mod05_ks01"])
def test_oracle_backward_pinned_to_reference_on_posed_scenes(name):
    """The oracle's backward against the reference's own source under a posed camera and scale_modifier != 1: blend gradients
    1e-4, the per-Gaussian stage (computeView2Gaussian_backward through R_view * R_q, backward.cu:381-587; SH backward with the
    world-space direction, :20-139) 1e-5 on the reference's own dL_dview2gaussian / dL_dcolors."""
    from test_parity_gpu import SCENES
    sc = SCENES[name]()
    o, ref = _pin_forward(sc)
    dL = np.random.default_rng(5).normal(size=(9, sc["H"], sc["W"])).astype(np.float32)
    gr, go = ref.backward(dL), o.backward(dL)
    for k in ("means2D", "colors", "opacity", "view2gaussian"):
        assert np.abs(gr[k].reshape(go[k].shape) - go[k]).max() <= 1e-4 * np.abs(go[k]).max(), k
    iso = o.preprocess_backward(gr["view2gaussian"], gr["colors"])
    for k in ("means3D", "sh", "scales", "rotations"):
        ref_v = gr[k].reshape(iso[k].shape)
        assert np.abs(ref_v - iso[k]).max() <= 1e-5 * max(np.abs(iso[k]).max(), 1e-20), k


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
@pytest.mark.parametrize("ks", [0.0, 0.1])
def test_oracle_pinned_to_reference_sh_degrees_and_kernel_sizes(deg, ks):
    sc = S.scene_frustum(5000, W=160, H=112, focal=120.0, seed=30 + deg, kernel_size=ks, sh_degree=deg)
    o, ref = _pin_forward(sc)
    dL = np.random.default_rng(deg).normal(size=(9, sc["H"], sc["W"])).astype(np.float32)
    gr, go = ref.backward(dL), o.backward(dL)
    for k in ("means2D", "colors", "opacity", "view2gaussian"):
        assert np.abs(gr[k].reshape(go[k].shape) - go[k]).max() <= 1e-4 * np.abs(go[k]).max(), k


def test_product_and_oracle_pinned_to_reference_at_s1m():
    """The headline configuration (1M Gaussians, 1600x1063, kernel_size 0.0) against the reference's own source: the PRODUCT's
    integer stages (radii, instance count, the 8.8M-entry sorted list, tile ranges) bit-exact, its image equal up to exp()."""
    sc = S.scene_frustum(1_000_000, seed=0)
    sd = to_dev(sc)
    ref = rb.Reference(sd, "_nofma")
    rc, rrad = ref.forward()
    res = product_forward_raw(sd)
    torch.cuda.synchronize()
    assert res["R"] == ref.R and ref.R > 8_000_000
    assert np.array_equal(res["radii"].cpu().numpy(), rrad)
    assert np.array_equal(fetch(res, "point_list").view(np.uint32), ref.fetch("point_list"))
    assert np.array_equal(fetch(res, "ranges").view(np.uint32).ravel(), ref.fetch("ranges").ravel())
    for name in ("depths", "means2D", "conic_opacity", "rgb", "view2gaussian"):
        vis = rrad > 0
        a = fetch(res, name).reshape(len(rrad), -1)[vis]; b = ref.fetch(name).reshape(len(rrad), -1)[vis]
        assert np.array_equal(bits(a), bits(b)), name
    pc = res["color"].cpu().numpy()
    d = np.abs(pc - rc)
    assert np.percentile(d, 99.9) <= 2e-5 * max(1.0, np.abs(rc).max()), np.percentile(d, [50, 99, 99.9, 100])
    ncp = fetch(res, "n_contrib").view(np.uint32).ravel(); ncr = ref.fetch("n_contrib").ravel()
    assert (ncp != ncr).mean() < 2e-3


def test_product_and_oracle_pinned_to_reference_at_s1m_posed():
    """The same at full size under a random rigid pose: the product's radii, instance count, sorted list, ranges and every K1 float
    (view2gaussian through R_view * R_q, the SH colour with the world-space view direction) bit-exact against the reference's own
    source; image equal up to exp()."""
    sc = S.scene_frustum(1_000_000, seed=0, pose_seed=0)
    sd = to_dev(sc)
    ref = rb.Reference(sd, "_nofma")
    rc, rrad = ref.forward()
    res = product_forward_raw(sd)
    torch.cuda.synchronize()
    assert res["R"] == ref.R and ref.R > 8_000_000
    assert np.array_equal(res["radii"].cpu().numpy(), rrad)
    assert np.array_equal(fetch(res, "point_list").view(np.uint32), ref.fetch("point_list"))
    assert np.array_equal(fetch(res, "ranges").view(np.uint32).ravel(), ref.fetch("ranges").ravel())
    for name in ("depths", "means2D", "conic_opacity", "rgb", "view2gaussian"):
        vis = rrad > 0
        a = fetch(res, name).reshape(len(rrad), -1)[vis]; b = ref.fetch(name).reshape(len(rrad), -1)[vis]
        assert np.array_equal(bits(a), bits(b)), name
    pc = res["color"].cpu().numpy()
    d = np.abs(pc - rc)
    assert np.percentile(d, 99.9) <= 2e-5 * max(1.0, np.abs(rc).max()), np.percentile(d, [50, 99, 99.9, 100])


def test_reference_truncates_contributor_ids_to_uint16_and_the_oracle_follows_it():
    """Tile lists beyond 65535 entries: the reference stores a pixel's contributor positions as uint16 (forward.cu:879, 983) and its
    point pass then evaluates the entries at position mod 65536 (forward.cu:1145).  The oracle restates that, the product
    reproduces it (test_parity_gpu.py::test_integrate_reproduces_the_uint16...); here the oracle is held against the reference
    itself on the same 78 000-entry list."""
    from test_parity_gpu import uint16_scene
    sc = uint16_scene()
    pts = np.ascontiguousarray(S.tetra_points(sc)[::40], dtype=np.float32)
    ref = rb.Reference(to_dev(sc), "_nofma")
    rc, ral, rcol, rrad = ref.integrate(pts)
    o = ob.OracleScene(sc)
    oc, oal, ocol, orad = o.integrate(pts)
    assert np.array_equal(rrad, orad) and np.array_equal(rc[8], oc[8])
    last = o.fetch("n_contrib").reshape(2, 16, 16)[0]
    assert (last > 65535).sum() >= 20
    d = np.abs(ral - oal)
    assert np.percentile(d, 99) < 1e-5 and d.max() < 5e-2, (np.percentile(d, [50, 99, 99.9]), d.max())


# ---- end-to-end PARAMETER gradients (dL_dmeans3D / dL_dscales / dL_drotations / dL_dsh): the per-Gaussian backward is an
# ---- ill-conditioned function of the atomically accumulated dL_dview2gaussian, so two runs of the REFERENCE ITSELF differ; that
# ---- spread -- measured here on the reference's own source -- is the yardstick the product is held to ----

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "product_param_grad_errors.json")) as _f:
    RECORDED = json.load(_f)["errors"]


def _rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


@pytest.mark.parametrize("name", ["small_ks01", "lego10k", "ragged", "long_lists", "mid100k", "posed_ragged", "posed_mid100k", "posed_mod2",
                                  "posed_mod05_ks01", "posed_stress_box"])
def test_parameter_gradients_within_twice_the_references_own_run_to_run_band(name):
    """product vs oracle (double accumulation in list order = the noise-free value of the reference's formulas) for the gradients
    of the Gaussian PARAMETERS, end to end (blend backward + per-Gaussian backward): relative L2 error <= 2 x the error of the
    reference's own runs against the same oracle (its fp32 atomics in scheduling order), and <= 1e-4 wherever the reference itself
    achieves that."""
    from test_parity_gpu import SCENES, _product_backward
    sc = SCENES[name]()
    sd = to_dev(sc)
    o = ob.OracleScene(sc)
    oc, orad = o.forward()
    dL = np.random.default_rng(17).normal(size=oc.shape).astype(np.float32)
    go = o.backward(dL)
    ref = rb.Reference(sd, "_nofma")
    ref.forward()
    # six runs of the reference (three until round 5): its error against the noise-free value is the MAXIMUM over its runs, and a run's
    # error is bimodal on the ill-conditioned scenes (the reference's own runs differ by 0.7 % ... 6.6 % in dL_dscales here) -- with three
    # runs the yardstick occasionally came out at its low mode and failed a product whose own error never moves (it has no atomics)
    runs = [ref.backward(dL) for _ in range(6)]
    res = product_forward_raw(sd)
    gp = _product_backward(res, dL)
    report = {}
    for k in ("means3D", "scales", "rotations", "sh", "opacity"):
        ref_err = max(_rel_l2(r[k].reshape(go[k].shape), go[k]) for r in runs)        # the reference against the noise-free value
        ref_spread = max(_rel_l2(runs[i][k], runs[0][k]) for i in range(1, len(runs)))   # ... and against itself
        mine = _rel_l2(gp[k].reshape(go[k].shape), go[k])
        report[k] = (mine, ref_err, ref_spread)
        assert mine <= 2.0 * max(ref_err, ref_spread) + 1e-6, (k, report)
        if ref_err < 1e-4:
            assert mine < 1e-4, (k, report)
        # ... and to its OWN recorded error: the product has no atomics, its error against the oracle is a fixed number per scene and
        # tensor (tests/golden/product_param_grad_errors.json, generated by tests/golden/make_param_grad_errors.py) -- the reference's band
        # above is set by the worse of its two modes, in which a real regression of the product could hide (ADVICE, round 5)
        assert mine <= 1.5 * RECORDED[name][k] + 1e-9, (k, mine, RECORDED[name][k])
    # the measured figures of all scenes (incl. S1M) are committed: profiles/r03_parity_report.{json,md} (tests/devtools/dev_parity_report.py)
    print("parameter-gradient errors (product, reference, reference run-to-run):", report)
