"""CPU parity tests of the HIP kernels' SOURCE: gaussian-opacity-fields_amd/csrc/*.hip compiled for the host against
tests/hipemu/include/hip/hip_runtime.h (workgroups as fibers in wave64 lock step, cross-lane operations with the hardware's lane
semantics) and run through the same C ABI as the product -- compared with the oracle exactly like tests/test_parity_gpu.py does on
the GPU, with the same bars (forward bit-exact, blend gradients to the round-3 measurements, K9 on identical inputs).

What this covers without a GPU: the kernels' logic -- binning (radix sorts incl. decoupled look-back, scans, instance emission,
ranges), the tile scheduler, forward / backward blend, the per-Gaussian backward, the opacity-field query -- and every index they
compute.  What it cannot: the code hipcc generates for gfx950, v_rcp / v_rsq / v_exp (host stand-ins, used only under a
tolerance), timing.  The GPU suite stays the parity gate; this one catches logic errors where no GPU is at hand.
The emulated library is test infrastructure: the product never loads it (tests/test_host_api.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
import oracle_binding as ob  # noqa: E402
import synthetic_scenes as S  # noqa: E402
import test_parity_gpu as TP  # noqa: E402
from gpu_common import assert_fast_mode_matches_exact, bits  # noqa: E402

build_emu = pytest.importorskip("build_emu")
if not os.path.exists(build_emu.CXX):
    pytest.skip("no host clang++ (%s) to build the emulated library" % build_emu.CXX, allow_module_level=True)
import emu_binding as E  # noqa: E402
if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu", "_build", "BUILD_FAILED_build_emu")):
    raise RuntimeError("__graft_entry__.build() recorded a failed build_emu run (tests/hipemu/_build/BUILD_FAILED_build_emu): these tests must not be skipped over it -- fix the host build and run build() again")

@pytest.fixture(autouse=True)
def _no_write_past_a_workspace():
    """every workspace the tests hand to the library is allocated at EXACTLY the size its gof_*_bytes() states, with guard bytes behind"""
    yield
    assert E.guards_intact() == []


FAST = ["tiny", "one", "small_ks0", "small_ks01", "long_lists", "stress_box", "posed_tiny", "posed_small_ks01", "posed_long_lists",
        "posed_stress_box", "posed_mod2", "posed_mod05_ks01", "emit_edges", "sub_tile", "strip_h", "strip_v", "one_px"]
MEDIUM = ["lego10k", "posed_ragged", "posed_clustered150k"]


def _pair(sc, **over):
    """oracle + the emulated library in BOTH forward modes: `e` = the default mode (the exact arithmetic without its fp64 divisions, pair_nodiv_cc: what ships; the
    integer arrays and the backward are checked on it), `e.exact` = the verification mode, whose image `pc` is held to the oracle's
    bits; the two modes are compared on the way (same decisions, floats within gpu_common.FAST_MODE_TOL)."""
    o = ob.OracleScene(sc, **over)
    oc, orad = o.forward()
    ex = E.EmuScene(sc, exact=True, **over)
    pc, prad = ex.forward()
    e = E.EmuScene(sc, **over)
    pf, prf = e.forward()
    if e.R > 0:
        assert_fast_mode_matches_exact(e.mode_arrays(), ex.mode_arrays())
    e.exact_scene = ex
    return o, oc, orad, e, pc, prad


@pytest.mark.parametrize("name", FAST + MEDIUM)
def test_emulated_forward_bit_exact(name):
    sc = TP.SCENES[name]()
    o, oc, orad, e, pc, prad = _pair(sc)
    P = len(orad); vis = orad > 0
    assert np.array_equal(prad, orad)
    assert e.R == o.num_rendered()
    for arr in TP.K1_ARRAYS:
        a = e.fetch(arr).reshape(P, -1)[vis]; b = o.fetch(arr).reshape(P, -1)[vis]
        assert TP._same(a, b), arr
    for arr in TP.INT_ARRAYS:
        assert TP._same(e.fetch(arr), o.fetch(arr)), arr
    assert np.array_equal(bits(e.exact_scene.fetch("final_T")), bits(o.fetch("final_T")))
    TP.assert_image_matches(pc, oc)
    # the tile scheduler: every tile in exactly one XCD queue, all queues consumed
    T = ((sc["W"] + 15) // 16) * ((sc["H"] + 15) // 16)
    q = e.fetch("tile_queue"); order = e.fetch("tile_order").reshape(8, -1)
    lens = q[8:16]
    assert lens.sum() == T and np.array_equal(q[0:8] >= lens, np.ones(8, bool))
    seen = np.concatenate([order[x, :lens[x]] for x in range(8)])
    assert np.array_equal(np.sort(seen), np.arange(T))


@pytest.mark.parametrize("name", FAST + MEDIUM)
def test_emulated_backward(name):
    sc = TP.SCENES[name]()
    o, oc, orad, e, pc, prad = _pair(sc)
    dL = np.random.default_rng(1).normal(size=oc.shape).astype(np.float32)
    go = o.backward(dL)
    gp = e.backward(dL)
    for k in ("means2D", "colors", "opacity", "view2gaussian"):
        TP.assert_grad_close(gp[k], go[k], k)
    iso = o.preprocess_backward(gp["view2gaussian"], gp["colors"])
    for k in ("means3D", "sh", "scales", "rotations"):
        TP.assert_k9_close(gp[k], iso[k], k)
    inv = orad <= 0
    for k, v in gp.items():
        assert np.isfinite(v).all(), k                       # (the buffers were pre-filled with NaN: every element was written)
        assert not v.reshape(len(orad), -1)[inv].any(), k


def test_emulated_backward_is_bit_reproducible():
    """no atomics in the backward: two runs give identical bits whatever the order the workgroups ran in (8 OS threads here)"""
    sc = TP.SCENES["posed_ragged"]()
    e = E.EmuScene(sc)
    pc, _ = e.forward()
    dL = np.random.default_rng(2).normal(size=pc.shape).astype(np.float32)
    a = e.backward(dL); b = e.backward(dL)
    for k in a:
        assert np.array_equal(bits(a[k]), bits(b[k])), k


def test_emulated_precomputed_inputs_forward_and_backward():
    """colors_precomp / view2gaussian_precomp / cov3D_precomp (gaussian_renderer/__init__.py:67-96): forward bit-exact, the
    gradients handed back for them against the oracle"""
    sc = S.scene_frustum(3000, W=160, H=112, focal=120.0, seed=12, kernel_size=0.1, pose_seed=9)
    o0 = ob.OracleScene(sc); o0.forward()
    P = sc["means3D"].shape[0]
    over = {"colors_precomp": np.random.default_rng(3).uniform(0, 1, (P, 3)).astype(np.float32),
            "view2gaussian_precomp": o0.fetch("view2gaussian").reshape(P, 10).copy()}
    o, oc, orad, e, pc, prad = _pair(sc, **over)
    assert np.array_equal(prad, orad)
    TP.assert_image_matches(pc, oc)
    dL = np.random.default_rng(4).normal(size=oc.shape).astype(np.float32)
    go = o.backward(dL); gp = e.backward(dL)
    for k in ("means2D", "colors", "opacity", "view2gaussian"):
        TP.assert_grad_close(gp[k], go[k], k)


@pytest.mark.parametrize("name", ["small_ks0", "long_lists", "stress_box", "posed_small_ks01", "posed_stress_box", "posed_ragged", "sub_tile", "strip_h", "strip_v", "one_px"])
def test_emulated_integrate_bit_exact(name):
    sc = TP.SCENES[name]()
    pts = np.ascontiguousarray(S.tetra_points(sc), dtype=np.float32)
    if len(pts) > 60_000:
        pts = pts[np.random.default_rng(3).choice(len(pts), 60_000, replace=False)]
    o = ob.OracleScene(sc)
    oc, oal, ocol, orad = o.integrate(pts)
    e = E.EmuScene(sc)
    c, a, colp, rad = e.integrate(pts)
    assert np.array_equal(rad, orad)
    assert np.array_equal(bits(c), bits(oc)), [int((bits(c[i]) != bits(oc[i])).sum()) for i in range(9)]
    assert np.array_equal(bits(a), bits(oal)), (int((bits(a) != bits(oal)).sum()), np.abs(a - oal).max())
    assert np.array_equal(bits(colp), bits(ocol))


@pytest.mark.parametrize("which", ["at_the_cap", "below_the_cap"])
def test_emulated_integrate_uint16_wrap_and_the_contributor_cap(which):
    """lists beyond 65535 entries (the reference's uint16 contributor ids wrap, forward.cu:879, 983, 1145): a tile whose pixels meet the
    1024-contributor cap -- the ray-centric pixel pass abandons it to the pixel-centric kernel -- and one that stays below it (the wrap
    inside integrate_rays' assembly); both forms of the pixel pass, every output bit the oracle's"""
    sc = TP.uint16_scene() if which == "at_the_cap" else TP.uint16_scene_below_the_cap()
    pts = np.ascontiguousarray(S.tetra_points(sc)[::40], dtype=np.float32)
    o = ob.OracleScene(sc)
    oc, oal, ocol, orad = o.integrate(pts)
    assert o.num_rendered() > 70_000 and (o.fetch("n_contrib").reshape(2, 16, 16)[0] > 65535).sum() >= 20
    lib = E.load()
    for mode in (0, 1):
        prev = lib.gof_set_integrate_pixel_pass(mode)
        try:
            c, a, colp, rad = E.EmuScene(sc).integrate(pts)
        finally:
            lib.gof_set_integrate_pixel_pass(prev)
        assert np.array_equal(rad, orad)
        assert np.array_equal(bits(c), bits(oc)), (mode, [int((bits(c[i]) != bits(oc[i])).sum()) for i in range(9)])
        assert np.array_equal(bits(a), bits(oal)) and np.array_equal(bits(colp), bits(ocol)), mode


@pytest.mark.parametrize("name", ["long_lists", "posed_stress_box", "strip_v"])
def test_emulated_integrate_pixel_centric_form_gives_the_same_bits(name):
    sc = TP.SCENES[name]()
    pts = np.ascontiguousarray(S.tetra_points(sc)[:40_000], dtype=np.float32)
    lib = E.load()
    rays = E.EmuScene(sc).integrate(pts)
    prev = lib.gof_set_integrate_pixel_pass(1)
    try:
        pix = E.EmuScene(sc).integrate(pts)
    finally:
        lib.gof_set_integrate_pixel_pass(prev)
    for x, y in zip(rays, pix):
        assert np.array_equal(bits(x), bits(y))


def test_emulator_reports_divergent_cross_lane_use():
    """the emulator's own contract: lanes of one wave meeting at different cross-lane sites is an error it reports, not a silent
    mis-pairing (checked on the runtime directly: tests/hipemu/selftest.cpp)"""
    import subprocess
    out = os.path.join(build_emu.OUT, "selftest")
    os.makedirs(build_emu.OUT, exist_ok=True)
    subprocess.check_call([build_emu.CXX, "-std=c++17", "-O1", "-g", "-pthread", "-I", os.path.join(build_emu.HERE, "include"),
                           os.path.join(build_emu.HERE, "selftest.cpp"), os.path.join(build_emu.HERE, "hipemu_rt.cpp"), "-o", out])
    r = subprocess.run([out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all ok" in r.stdout
    r = subprocess.run([out, "diverge"], capture_output=True, text=True)
    assert r.returncode != 0 and "DIFFERENT cross-lane operations" in r.stderr


@pytest.mark.parametrize("seed", range(24))
def test_emulated_fuzz_forward_integrate_backward(seed):
    """the GPU suite's randomised small scenes (image size, focal length, splat size from sub-pixel to tile-covering, anisotropy,
    opacity regimes, depth range, kernel size, SH degree, pose, scale_modifier): forward and opacity query bit-exact, blend
    gradients within the GPU suite's bars"""
    sc = TP._fuzz_scene(seed)
    o, oc, orad, e, pc, prad = _pair(sc)
    assert e.R == o.num_rendered() and np.array_equal(prad, orad)
    for arr in TP.INT_ARRAYS:
        assert TP._same(e.fetch(arr), o.fetch(arr)), arr
    assert np.array_equal(bits(e.exact_scene.fetch("final_T")), bits(o.fetch("final_T")))
    TP.assert_image_matches(pc, oc)
    dL = np.random.default_rng(seed).normal(size=oc.shape).astype(np.float32)
    go = o.backward(dL); gp = e.backward(dL)
    for k in ("means2D", "colors", "opacity", "view2gaussian"):
        TP.assert_grad_close(gp[k], go[k], k)
    pts = np.ascontiguousarray(S.tetra_points(sc)[:20000], dtype=np.float32)
    io, ia, icol, _ = o.integrate(pts)
    c, a, colp, _ = e.integrate(pts)
    assert np.array_equal(bits(c), bits(io)) and np.array_equal(bits(a), bits(ia)) and np.array_equal(bits(colp), bits(icol))


def test_emulated_sync_free_forward_respects_its_capacity():
    """gof_forward_fused: every launch behind the scan is sized for the caller's capacity and reads the true count on the device.
    With room to spare and with EXACTLY the count it renders the two-stage forward's image; one instance short it returns
    GOF_E_CAPACITY -- and in every case nothing is written behind a binning workspace allocated at exactly its stated size."""
    sc = TP.SCENES["posed_mod2"]()
    e = E.EmuScene(sc)
    want, _ = e.forward()
    want = want.copy(); R = e.R
    list_two_stage = e.fetch("point_list").copy()
    for cap in (R + 12345, R):
        f = E.EmuScene(sc)
        rc, count, intact = f.forward_fused(cap)
        assert rc == 0 and count == R and intact, (cap, rc, count, intact)
        assert np.array_equal(bits(f.color), bits(want))
        assert np.array_equal(f.fetch("point_list")[:R], list_two_stage)
        g = f.backward(np.ones_like(want))                  # the backward finds its arrays through the capacity's layout
        assert all(np.isfinite(v).all() for v in g.values())
        # the frame's counters, stored into the caller's host words by the forward's last kernel, decode to what the synchronising query reports
        assert f.usage_decoded() == (f.staged, f.masks_requested, f.masks_held) and f.staged > 0
    for cap in (R - 1, R // 2, 64):
        f = E.EmuScene(sc)
        rc, count, intact = f.forward_fused(cap)
        assert rc == -5 and count == R and intact, (cap, rc, count, intact)          # GOF_E_CAPACITY, the true count reported


def test_emulated_compact_workspaces_the_mask_pool_and_the_record_pool():
    """Round 4: the contributor masks live in a POOL of sub-chunks whose size the caller chooses (gof_binning_bytes_for) and the
    backward's partial gradient records in a pool sized from what the forward staged (gof_backward_query +
    gof_backward_scratch_bytes_for).  With both pools at EXACTLY the reported need (guard bytes behind every workspace) the gradients
    are bit-identical to the worst-case workspaces'; a mask pool one sub-chunk short is reported by the query (requested > held)
    before any backward runs, and the image of that forward is still exact."""
    exercised = 0
    for name in ("posed_ragged", "posed_long_lists", "lego10k"):
        sc = TP.SCENES[name]()
        full = E.EmuScene(sc)
        img, _ = full.forward()
        dL = np.random.default_rng(11).normal(size=img.shape).astype(np.float32)
        want = full.backward(dL, full_scratch=True)
        again = full.backward(dL)                                   # record pool at exactly the staged count
        need, staged = full.masks_requested, full.staged
        assert 0 < staged <= full.R and 0 < need <= full.masks_held
        T = ((sc["W"] + 15) // 16) * ((sc["H"] + 15) // 16)
        assert full.masks_held == 4 * (full.R // 256 + T + 2 + 64)   # the worst case: a chunk of 4 sub-chunks per (tile, batch) + the shards' rounding
        for k in want:
            assert np.array_equal(bits(want[k]), bits(again[k])), (name, k)
        tight = E.EmuScene(sc)
        slack = 4 * 64                                              # (a pool of 64 shards serves 64 * floor(chunks / 64) chunks)
        img2, _ = tight.forward(mask_subchunks=need + slack)        # mask pool at the need
        assert np.array_equal(bits(img2), bits(img))
        got = tight.backward(dL)
        # (the pool lies on top of the tile sort's two ping-pong buffers, dead by then: a workspace never holds fewer sub-chunks than those 8 B per instance make)
        floor = E.EmuScene(sc); floor.forward(mask_subchunks=0)
        try:
            floor.backward(dL)
        except RuntimeError:
            pass
        assert tight.masks_held == max(need + slack, floor.masks_held) and tight.masks_requested == need and need % 4 == 0
        for k in want:
            assert np.array_equal(bits(want[k]), bits(got[k])), (name, k)
        assert tight.binning.size < full.binning.size
        if need - 4 > floor.masks_held:
            exercised += 1
            short = E.EmuScene(sc)
            img3, _ = short.forward(mask_subchunks=need - 4)        # one chunk (4 sub-chunks: a tile's four waves x one batch) short
            assert np.array_equal(bits(img3), bits(img))            # the image does not depend on the pool
            with pytest.raises(RuntimeError, match="mask pool too small"):
                short.backward(dL)
            assert short.masks_requested > short.masks_held == need - 4
    assert exercised >= 2


def test_emulated_cull_audit_counts_no_dropped_pair():
    """the instrumented build (-DGOF_STATS -DGOF_CULL_AUDIT: the consumption walks EVERY list entry and counts the pairs the exact
    path accepts that the footprint-conic scan had not marked) run from source on the host: 0 dropped on the stress scenes, the
    table's posed scenes and 24 fuzz seeds -- the GPU suite's audit (test_the_cull_scan_drops_no_pair_the_exact_path_accepts)
    without a GPU, at the sizes the CPU affords"""
    import ctypes as C
    lib = E.load(extra_flags=("-DGOF_STATS", "-DGOF_CULL_AUDIT"), tag="audit")
    out = (C.c_ulonglong * 8)()
    scenes = [TP.SCENES[k]() for k in ("stress_box", "posed_stress_box", "long_lists", "posed_long_lists", "posed_mod2", "posed_mod05_ks01", "small_ks01", "posed_ragged")]
    scenes += [TP._fuzz_scene(s) for s in range(24)]
    scenes.append(S.scene_frustum(40_000, W=320, H=208, focal=240.0, seed=7, sigma_px=0.4, zmin=5.0, zmax=80.0, pose_seed=9))     # far sub-pixel splats
    total = total_int = 0
    for sc in scenes:
        lib.gof_debug_fw_stats(out, 1)
        e = E.EmuScene(sc, lib=lib, exact=True)              # (the pairs the EXACT arithmetic accepts)
        e.forward()
        lib.gof_debug_fw_stats(out, 1)
        s = list(out)
        assert s[6] == 0, ("pairs dropped by the cull scan", s[6], s[3])
        total += s[3]
        # round 5: the ray-centric pixel pass of the opacity-field query (integrate_rays: the conic at the lane's own ray)
        iout = (C.c_ulonglong * 16)()
        lib.gof_debug_int_stats(iout, 1)
        e.integrate(np.ascontiguousarray(sc["means3D"][:500], dtype=np.float32))
        lib.gof_debug_int_stats(iout, 1)
        assert iout[11] == 0, ("(ray, entry) pairs dropped by integrate_rays' scan", iout[11], iout[7])
        total_int += iout[7]
    assert total > 2_000_000 and total_int > 2_000_000          # accepted pairs examined


@pytest.mark.parametrize("order", ["reverse", "random:7"])
def test_results_do_not_depend_on_the_order_the_emulator_runs_fibers_in(order):
    """The hardware fixes no order between the waves of a workgroup and runs all lanes of an instruction together, so between two
    synchronisation points no result may depend on the order the emulator happens to run fibers in (ascending by default).  The
    same parity tests under a descending and a pseudo-random order (HIPEMU_ORDER, read once per process: a subprocess): a failure
    here is a missing barrier between waves or an unlisted lock-step point inside one (build_emu.py: LOCKSTEP_POINTS)."""
    import subprocess
    sel = "posed_small_ks01 or posed_stress_box or posed_long_lists or posed_mod2 or small_ks0 or precomputed or fuzz_forward_integrate_backward"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "(%s) and not order" % sel, "-p", "no:cacheprovider"],
                       env=dict(os.environ, HIPEMU_ORDER=order), capture_output=True, text=True, timeout=1500,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_emulated_compressed_sh_gradient_exchange_equals_the_dense_sum():
    """Data-parallel exchange of the SH gradient in compressed form (include/gof_hip.h: gof_sh_grad_pack / gof_sh_grad_expand;
    dp/reducer.py): dL_dsh[k] = basis_k(direction) * dL_dRGB (masked by the forward's clamp flags), so a rank sends 12 B per Gaussian
    -- its colour gradient -- plus the camera centre, and every rank expands the sum over the views locally, in rank order.  With
    the real kernels (run from source on the host): the expansion of two views' packed gradients equals the sum of the two views'
    dense dL_dsh from the backward -- every element the same number (a zero may carry the other sign: basis x 0) -- for the
    [P,16,3] layout and for the split (dc, rest) layout; scale = 1/2 averages."""
    import ctypes as C
    lib = E.load()
    base = S.scene_frustum(2500, W=128, H=96, focal=100.0, seed=31, kernel_size=0.1)
    views = [base, S.other_view(base, 1)]                       # one cloud, two cameras
    P, M = base["means3D"].shape[0], base["shs"].shape[1]
    dense, packed = [], np.zeros((2, P + 1, 3), np.float32)
    for v, sc in enumerate(views):
        e = E.EmuScene(sc)
        color, radii = e.forward()
        g = e.backward(np.random.default_rng(7 + v).normal(size=color.shape).astype(np.float32))
        dense.append(g["sh"].copy())
        rc = lib.gof_sh_grad_pack(P, E._p(g["colors"]), E._p(e.geom), e.geom.size, E._p(radii), E._p(packed[v]), None)
        assert rc == 0, lib.gof_last_error()
        packed[v, P] = sc["campos"]
    assert (dense[0] != 0).any() and (dense[1] != 0).any()
    want = dense[0] + dense[1]
    means = np.ascontiguousarray(base["means3D"], np.float32)
    vs = (P + 1) * 3
    base_ptr = packed.ctypes.data
    # one [P, M, 3] tensor
    out = np.full((P, M, 3), np.nan, np.float32)
    rc = lib.gof_sh_grad_expand(P, int(base["sh_degree"]), M, 2, E._p(means), C.c_void_p(base_ptr + 12 * P), vs, C.c_void_p(base_ptr), vs, 1.0,
                                C.c_void_p(out.ctypes.data), 3 * M, C.c_void_p(out.ctypes.data + 12), 3 * M, None)
    assert rc == 0, lib.gof_last_error()
    assert np.array_equal(out, want) and not (want[bits(out) != bits(want)] != 0).any()
    # split layout (features_dc [P,1,3], features_rest [P,15,3]) and the averaging scale
    dc = np.full((P, 1, 3), np.nan, np.float32); rest = np.full((P, M - 1, 3), np.nan, np.float32)
    rc = lib.gof_sh_grad_expand(P, int(base["sh_degree"]), M, 2, E._p(means), C.c_void_p(base_ptr + 12 * P), vs, C.c_void_p(base_ptr), vs, 0.5,
                                C.c_void_p(dc.ctypes.data), 3, C.c_void_p(rest.ctypes.data), 3 * (M - 1), None)
    assert rc == 0, lib.gof_last_error()
    got = np.concatenate([dc, rest], 1)
    assert np.array_equal(got, (0.5 * want).astype(np.float32))      # (a power-of-two scale commutes with the rounding)


def test_emulated_training_loop_fits_a_target_image():
    """gradient USEFULNESS end to end, without a GPU: a perturbed copy of a small scene is optimised (torch Adam on the host; the
    rasterizer's forward and backward are the kernels run from source) towards the image the unperturbed scene renders -- the L1
    loss must fall by more than half in 60 steps.  Wrong signs / scalings / index mix-ups in any gradient make it stall or rise."""
    import torch
    rng = np.random.default_rng(3)
    truth = S.scene_frustum(400, W=96, H=64, focal=80.0, seed=21, kernel_size=0.1, sigma_px=4.0, pose_seed=5)
    target, _ = E.EmuScene(truth).forward()
    target = target[:3].copy()
    P = truth["means3D"].shape[0]
    start = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in truth.items()}
    start["means3D"] = (truth["means3D"] + rng.normal(0, 0.02, (P, 3)) * np.abs(truth["means3D"][:, 2:3])).astype(np.float32)
    start["opacities"] = np.clip(truth["opacities"] * rng.uniform(0.5, 1.0, (P, 1)), 0.01, 0.99).astype(np.float32)
    start["shs"] = (truth["shs"] + rng.normal(0, 0.15, truth["shs"].shape)).astype(np.float32)
    start["scales"] = (truth["scales"] * np.exp(rng.normal(0, 0.15, (P, 3)))).astype(np.float32)
    names = {"means3D": "means3D", "opacities": "opacity", "shs": "sh", "scales": "scales", "rotations": "rotations"}
    params = {k: torch.from_numpy(start[k].copy()).requires_grad_(True) for k in names}
    lrs = {"means3D": 2e-3, "opacities": 2e-2, "shs": 1e-2, "scales": 1e-3, "rotations": 1e-3}
    opt = torch.optim.Adam([{"params": [params[k]], "lr": lrs[k]} for k in names], eps=1e-15)
    losses = []
    for it in range(60):
        sc = dict(start)
        for k in names:
            sc[k] = params[k].detach().numpy()
        sc["opacities"] = np.clip(sc["opacities"], 1e-3, 0.999)
        e = E.EmuScene(sc)
        img, _ = e.forward()
        diff = img[:3] - target
        losses.append(float(np.abs(diff).mean()))
        dL = np.zeros_like(img)
        dL[:3] = np.sign(diff) / diff.size                        # d mean|img - target| / d img
        g = e.backward(dL)
        opt.zero_grad()
        for k, gk in names.items():
            params[k].grad = torch.from_numpy(g[gk].reshape(params[k].shape).copy())
        opt.step()
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert min(losses[-10:]) < min(losses[:10])


def test_emulated_mesh_extraction_paths_min_over_views_and_packed_cache():
    """SURVEY 8(f)1, the mesh-extraction driver fusion, with the kernels run from source: (a) the point pass reading the PACKED
    per-view cache (68 B per Gaussian) gives the bits of the plain one; (b) the reduction over views of reference
    extract_mesh.py:17-34 -- color = where(alpha < alpha_min, color_view, color); alpha_min = min(alpha_min, alpha) -- fused into the
    point pass's store equals that reduction done on the separate per-view outputs, bit for bit, with and without the colour."""
    base = S.scene_frustum(3000, W=128, H=96, focal=100.0, seed=17, kernel_size=0.1, pose_seed=3)
    views = [base, S.other_view(base, 1), S.other_view(base, 2)]
    pts = np.ascontiguousarray(S.tetra_points(base)[::2], dtype=np.float32)
    PN = len(pts)
    want_alpha = np.ones(PN, np.float32); want_color = np.zeros((PN, 3), np.float32)
    acc_alpha = np.ones(PN, np.float32); acc_color = np.zeros((PN, 3), np.float32)
    acc_alpha_only = np.ones(PN, np.float32)
    for k, sc in enumerate(views):
        e = E.EmuScene(sc)
        e.integrate_view()
        out, alpha, colp = e.integrate_points(pts, "plain")
        e.pack_geom()
        out_p, alpha_p, colp_p = e.integrate_points(pts, "packed")
        assert np.array_equal(bits(out), bits(out_p)) and np.array_equal(bits(alpha), bits(alpha_p)) and np.array_equal(bits(colp), bits(colp_p))
        take = alpha < want_alpha                                       # extract_mesh.py:26-29
        want_color[take] = colp[take]
        want_alpha = np.minimum(want_alpha, alpha)
        e.integrate_points(pts, "min" if k % 2 == 0 else "min_packed", acc_alpha, acc_color)
        e.integrate_points(pts, "min_packed" if k % 2 == 0 else "min", acc_alpha_only, None)
    assert (want_alpha < 1).mean() > 0.5
    assert np.array_equal(bits(acc_alpha), bits(want_alpha)) and np.array_equal(bits(acc_color), bits(want_color))
    assert np.array_equal(bits(acc_alpha_only), bits(want_alpha))


def test_emulated_mark_visible_matches_oracle():
    lib = E.load()
    sc = S.scene_frustum(5000, W=160, H=112, focal=120.0, seed=4, pose_seed=6)
    sc["means3D"][::7, 2] *= -1.0
    m = np.ascontiguousarray(sc["means3D"], np.float32); V = np.ascontiguousarray(sc["viewmatrix"], np.float32); Pm = np.ascontiguousarray(sc["projmatrix"], np.float32)
    present = np.zeros(len(m), np.uint8)
    assert lib.gof_mark_visible(len(m), E._p(m), E._p(V), E._p(Pm), E._p(present), None) == 0
    want = ob.mark_visible(m, V, Pm)
    assert np.array_equal(present.astype(bool), np.asarray(want).astype(bool)) and 0 < present.sum() < len(m)


def test_emulated_staged_entry_points_equal_the_one_call_forms():
    """gof_backward_blend + gof_backward_preprocess (the split the data-parallel reducer uses to start its exchange early) give the
    bits of gof_backward; gof_integrate_run (the reference's monolithic IntegrateGaussiansToPointsCUDA shape) gives the bits of
    gof_integrate_view + gof_integrate_points; gof_profile_enable / gof_profile_report name the kernels of a call."""
    import ctypes as C
    import json
    lib = E.load()
    sc = TP.SCENES["posed_mod2"]()
    e = E.EmuScene(sc)
    assert lib.gof_profile_enable(1) == 0
    img, radii = e.forward()
    buf = C.create_string_buffer(1 << 16)
    assert lib.gof_profile_report(buf, len(buf)) == 0
    rep = json.loads(buf.value.decode())
    assert {"preprocess_fwd", "blend_forward"} <= set(rep) and all(v["calls"] >= 1 for v in rep.values())
    assert lib.gof_profile_enable(0) == 0
    dL = np.random.default_rng(3).normal(size=img.shape).astype(np.float32)
    one = e.backward(dL)
    # the same call, staged
    P, M = e.P, e.M
    g = {k: np.full_like(v, np.nan) for k, v in one.items()}
    g["cov3D"].fill(0)
    nscratch = lib.gof_backward_scratch_bytes(P, e.R)
    scratch = E._aligned(nscratch, what="backward scratch")
    call = (C.byref(e.args), e.R, E._p(e.radii), E._p(e.geom), e.geom.size, E._p(e.binning), e.binning.size, E._p(e.img), e.img.size, E._p(dL),
            E._p(g["means2D"]), E._p(g["colors"]), E._p(g["opacity"]), E._p(g["means3D"]), None, E._p(g["sh"]), None, E._p(g["scales"]), E._p(g["rotations"]),
            E._p(g["view2gaussian"]), E._p(scratch), nscratch, None)
    assert lib.gof_backward_blend(*call) == 0
    assert np.array_equal(bits(g["colors"]), bits(one["colors"])) and np.array_equal(bits(g["view2gaussian"]), bits(one["view2gaussian"]))   # final after stage 1
    assert lib.gof_backward_preprocess(*call) == 0
    for k in one:
        assert np.array_equal(bits(g[k]), bits(one[k])), k
    # integrate: monolithic vs split
    pts = np.ascontiguousarray(S.tetra_points(sc)[::3], dtype=np.float32); PN = len(pts)
    want_img, want_alpha, want_col, _ = E.EmuScene(sc).integrate(pts)
    f = E.EmuScene(sc)
    f.geom = E._aligned(lib.gof_geom_bytes(f.P), what="geom"); f.img = E._aligned(lib.gof_image_bytes(f.W, f.H), what="image")
    f.radii = np.zeros(f.P, np.int32)
    n = C.c_uint32(0)
    assert lib.gof_forward_prepare(C.byref(f.args), E._p(f.geom), f.geom.size, E._p(f.img), f.img.size, E._p(f.radii), C.byref(n), None) == 0
    pws = E._aligned(lib.gof_point_bytes(PN), what="point ws"); ni = C.c_uint32(0)
    assert lib.gof_integrate_prepare_points(C.byref(f.args), PN, E._p(pts), E._p(pws), pws.size, C.byref(ni), None) == 0
    binning = E._aligned(lib.gof_binning_bytes(n.value, f.W, f.H), what="binning"); pbin = E._aligned(lib.gof_point_binning_bytes(ni.value, f.W, f.H), what="point binning")
    out = np.zeros((9, f.H, f.W), np.float32); alpha = np.ones(PN, np.float32); col = np.zeros((PN, 3), np.float32)
    rc = lib.gof_integrate_run(C.byref(f.args), n.value, E._p(f.radii), PN, ni.value, E._p(f.geom), f.geom.size, E._p(binning), binning.size, E._p(f.img), f.img.size,
                               E._p(pws), pws.size, E._p(pbin), pbin.size, E._p(out), E._p(alpha), E._p(col), None)
    assert rc == 0, lib.gof_last_error()
    assert np.array_equal(bits(out), bits(want_img)) and np.array_equal(bits(alpha), bits(want_alpha)) and np.array_equal(bits(col), bits(want_col))


def test_emulated_tight_tile_rectangles_variant_changes_no_output():
    """gof_set_tight_tile_rects(1) (opt-in; the default keeps the reference's tile lists entry for entry): a Gaussian is
    binned only into the tiles of its 3-sigma square (auxiliary.h:64-74) that its footprint box -- widened by the pixel the opacity
    query's corner sub-rays need -- can reach.  Lists shrink (S1M: 8.84 M -> 6.99 M instances, 505 -> 439 entries scanned per pixel);
    image, state, radii and the opacity query stay bit-identical, gradients equal up to the summation order of the per-Gaussian gather."""
    scenes = [TP.SCENES[k]() for k in ("small_ks01", "posed_mod2", "posed_stress_box", "posed_long_lists", "posed_ragged")] + [TP._fuzz_scene(s) for s in (0, 9, 11, 13, 17, 21, 26, 29, 31, 38)]
    shrunk = 0
    for sc in scenes:
        e0 = E.EmuScene(sc); c0, r0 = e0.forward()
        e1 = E.EmuScene(sc, tight=True); c1, r1 = e1.forward()
        assert np.array_equal(bits(c0), bits(c1)) and np.array_equal(r0, r1) and np.array_equal(bits(e0.fetch("final_T")), bits(e1.fetch("final_T")))
        assert e1.R <= e0.R
        shrunk += e1.R < e0.R
        dL = np.random.default_rng(5).normal(size=c0.shape).astype(np.float32)
        g0, g1 = e0.backward(dL), e1.backward(dL)
        for k in ("means2D", "colors", "opacity", "view2gaussian", "sh"):
            assert np.abs(g0[k] - g1[k]).max() <= 2e-6 * max(np.abs(g0[k]).max(), 1e-30), k
        pts = np.ascontiguousarray(S.tetra_points(sc)[:20000], dtype=np.float32)
        for a, b in zip(E.EmuScene(sc).integrate(pts)[:3], E.EmuScene(sc, tight=True).integrate(pts)[:3]):
            assert np.array_equal(bits(a), bits(b))
    assert shrunk >= len(scenes) - 1


@pytest.mark.parametrize("name", ["lego10k", "posed_ragged", "posed_clustered150k"])
def test_emulated_single_kernel_radix_passes_with_long_look_back_chains(name):
    """os_pass (radix.hip) requests OS_LOOKBACK predecessor descriptors per look-back round trip and consumes them in order.  The default
    build runs few tiles per pass at these scene sizes; this variant sorts in tiles of 256 items and sends EVERY sort -- depth, tile and
    the query points' -- through the single-kernel passes (hundreds to thousands of tiles per pass, several look-back rounds per tile),
    and the fused rectangle gather + scan (binning.hip: gather_scan_rects, 64 predecessors per round trip) in tiles of 256 positions:
    lists, keys, ranges and the opacity query must stay the oracle's bits."""
    lib = E.load(extra_flags=("-DGOF_RS_CHUNK=64", "-DGOF_OS_MAX_UNITS=1048576", "-DGOF_GS_ITEMS=1"), tag="chains")
    sc = TP.SCENES[name]()
    o = ob.OracleScene(sc)
    oc, orad = o.forward()
    e = E.EmuScene(sc, lib=lib)
    pc, prad = e.forward()
    assert np.array_equal(prad, orad) and e.R == o.num_rendered()
    for arr in TP.INT_ARRAYS:
        assert TP._same(e.fetch(arr), o.fetch(arr)), arr
    pts = np.ascontiguousarray(S.tetra_points(sc)[:30000], dtype=np.float32)
    for a, b in zip(E.EmuScene(sc, lib=lib).integrate(pts)[:3], E.EmuScene(sc).integrate(pts)[:3]):
        assert np.array_equal(bits(a), bits(b))


@pytest.mark.parametrize("name", ["posed_mod2", "lego10k", "posed_ragged"])
def test_emulated_capacity_far_above_the_count_in_the_histogram_scan_scatter_sort(name):
    """gof_forward_fused with a capacity of 3.5x / 1.01x the instance count and every sort as histogram / scan / scatter launches (build
    variant GOF_OS_MAX_UNITS=0): only the blocks that hold items take part -- they are the histogram's stride and the bound of its scan
    (radix.hip: rs_active_blocks), tile_ranges strides over the count -- and the lists are those of the two-stage forward, nothing
    written behind the workspaces.  The sort's blocks are the emission's workgroups here as in the shipped build beyond 2 M instances:
    emit_instances leaves the first pass's histogram (binning.hip: hist0), on both forwards."""
    lib = E.load(extra_flags=("-DGOF_OS_MAX_UNITS=0",), tag="classic4k")
    sc = TP.SCENES[name]()
    v = E.EmuScene(sc, lib=lib); v.forward()          # the two-stage forward of the variant (host-known count) against the shipped build's
    d = E.EmuScene(sc); d.forward()
    assert v.R == d.R and np.array_equal(v.fetch("point_list"), d.fetch("point_list")) and np.array_equal(v.fetch("ranges"), d.fetch("ranges"))
    e = E.EmuScene(sc)
    want, _ = e.forward()
    want = want.copy(); R = e.R
    list_two_stage = e.fetch("point_list").copy()
    ranges = e.fetch("ranges").copy()
    for cap in (int(3.5 * R) + 777, R + R // 100 + 1, R):
        f = E.EmuScene(sc, lib=lib)
        rc, count, intact = f.forward_fused(cap)
        assert rc == 0 and count == R and intact, (cap, rc, count, intact)
        assert np.array_equal(bits(f.color), bits(want))
        assert np.array_equal(f.fetch("point_list")[:R], list_two_stage) and np.array_equal(f.fetch("ranges"), ranges)



