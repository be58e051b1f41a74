"""CPU tests of the kernels behind the SURVEY 8(f) rows -- training epilogue (SSIM, L1, depth -> normal, the one-call loss, Adam),
simple_knn.distCUDA2, marching tetrahedra -- with their SOURCE compiled for the host (tests/hipemu) and compared with the same
oracles as the GPU tests (tests/test_train_epilogue_gpu.py, tests/test_knn.py, tests/test_mtets_gpu.py).  Test infrastructure: the
Python mirrors refuse host tensors; the fixture below swaps the emulated library in for THIS module's calls only."""
import contextlib
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "hipemu"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import oracle_binding as ob  # noqa: E402
import synthetic_scenes as S  # noqa: E402
import knn_oracle as KO  # noqa: E402
import train_epilogue_oracle as O  # noqa: E402

build_emu = pytest.importorskip("build_emu")
if not os.path.exists(build_emu.CXX):
    pytest.skip("no host clang++ (%s) to build the emulated library" % build_emu.CXX, allow_module_level=True)
import emu_binding as E  # noqa: E402


@pytest.fixture()
def emu(monkeypatch):
    """train_epilogue's ctypes layer pointed at the emulated library (signatures declared on it by the package's own table); the
    device checks of that layer accept host tensors for the duration of one test."""
    import train_epilogue._backend as TB
    lib = E.load()
    monkeypatch.setattr(TB, "lib", lib)
    TB._declare()
    monkeypatch.setattr(TB, "_stream", lambda: None)
    monkeypatch.setattr(TB, "_same_device", lambda *a: contextlib.nullcontext())

    def f32(t, what):
        assert t.dtype == torch.float32, what
        return t.contiguous()
    monkeypatch.setattr(TB, "_need_cuda_f32", f32)
    return lib


def _close(a, b, tol, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    m = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= tol * m, (what, np.abs(a - b).max() / m)


@pytest.mark.parametrize("shape", [(3, 101, 77), (3, 5, 300), (1, 1, 1), (2, 3, 40, 33)])
def test_emulated_ssim_matches_oracle(emu, shape):
    import train_epilogue as T
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(shape, generator=g)
    y = (x + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    xp = x.clone().requires_grad_(True)
    s = T.ssim(xp, y)
    (gx,) = torch.autograd.grad(s, xp)
    xo = x.clone().requires_grad_(True)
    so = O.ssim(xo, y)
    (go,) = torch.autograd.grad(so, xo)
    assert float(s.detach()) == pytest.approx(so.item(), rel=5e-6)
    _close(gx.numpy(), go.numpy(), 5e-5, "d ssim / d img1 %s" % (shape,))


class _Cam:
    def __init__(self, wvt, W, H, fovx, fovy):
        self.world_view_transform = wvt; self.image_width = W; self.image_height = H; self.FoVx = fovx; self.FoVy = fovy


@pytest.mark.parametrize("W,H", [(130, 67), (2, 9), (1, 1)])
def test_emulated_depth_to_normal_matches_oracle(emu, W, H):
    import train_epilogue as T
    g = torch.Generator().manual_seed(W * 7 + H)
    q = torch.randn(4, generator=g); q = q / q.norm()
    w_, x_, y_, z_ = q.tolist()
    R = torch.tensor([[1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - w_ * z_), 2 * (x_ * z_ + w_ * y_)],
                      [2 * (x_ * y_ + w_ * z_), 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - w_ * x_)],
                      [2 * (x_ * z_ - w_ * y_), 2 * (y_ * z_ + w_ * x_), 1 - 2 * (x_ * x_ + y_ * y_)]])
    M = torch.eye(4); M[:3, :3] = R; M[:3, 3] = torch.randn(3, generator=g)
    wvt = M.T.contiguous()
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    depth = (3.0 + 0.5 * torch.sin(xx / 9.0) + 0.3 * torch.cos(yy / 7.0) + 0.02 * torch.rand((H, W), generator=g))[None]
    wn = torch.randn((H, W, 3), generator=g); wp = torch.randn((H, W, 3), generator=g)
    dp = depth.clone().requires_grad_(True)
    n, p = T.depth_to_normal(_Cam(wvt, W, H, 0.9, 0.65), dp)
    (gd,) = torch.autograd.grad((n * wn).sum() + (p * wp).sum(), dp)
    do = depth.clone().requires_grad_(True)
    no, po = O.depth_to_normal(wvt, W, H, 0.9, 0.65, do)
    (go,) = torch.autograd.grad((no * wn).sum() + (po * wp).sum(), do)
    _close(p.detach().numpy(), po.detach().numpy(), 2e-6, "points")
    d64 = depth.double().requires_grad_(True)
    n64, p64 = O.depth_to_normal(wvt.double(), W, H, 0.9, 0.65, d64, dtype=torch.float64)
    (g64,) = torch.autograd.grad((n64 * wn.double()).sum() + (p64 * wp.double()).sum(), d64)
    n64, g64 = n64.detach().numpy(), g64.numpy()
    err_ref_n = np.abs(no.detach().numpy() - n64).max(); err_ref_g = np.abs(go.numpy() - g64).max()
    assert np.abs(n.detach().numpy() - n64).max() <= 2 * err_ref_n + 1e-6
    assert np.abs(gd.numpy() - g64).max() <= 2 * err_ref_g + 1e-6 * max(np.abs(g64).max(), 1e-30)


@pytest.mark.parametrize("shape", [(3, 37, 53), (1,), (8193,)])
def test_emulated_l1_matches_oracle(emu, shape):
    from train_epilogue.loss_utils import _L1
    g = torch.Generator().manual_seed(5)
    a = torch.randn(shape, generator=g); b = torch.randn(shape, generator=g)
    ap = a.clone().requires_grad_(True)
    v = _L1.apply(ap, b)
    (ga,) = torch.autograd.grad(v, ap)
    ao = a.clone().requires_grad_(True)
    vo = O.l1_loss(ao, b)
    (go,) = torch.autograd.grad(vo, ao)
    assert float(v.detach()) == pytest.approx(vo.item(), rel=2e-6)
    assert np.array_equal(ga.numpy(), go.numpy())


@pytest.mark.parametrize("W,H,lambdas", [(203, 131, (0.2, 0.05, 100.0)), (16, 16, (0.2, 0.0, 0.0)), (3, 3, (0.5, 1.0, 1.0))])
def test_emulated_training_loss_matches_oracle(emu, W, H, lambdas):
    """the one-call loss (gof_train_loss: L1 + SSIM + depth -> normal consistency + distortion, train.py:150-186) and its gradient
    w.r.t. the rendering, against the oracle's composition of the reference's expressions under torch autograd"""
    import train_epilogue._backend as TB
    g = torch.Generator().manual_seed(W + H)
    rend = torch.rand((9, H, W), generator=g)
    nrm = torch.randn((3, H, W), generator=g); rend[3:6] = nrm / nrm.norm(dim=0, keepdim=True) * torch.rand((1, H, W), generator=g)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    rend[6] = 3.0 + 0.5 * torch.sin(xx / 9.0) + 0.3 * torch.cos(yy / 7.0)
    rend[7] = 0.2 + 0.8 * torch.rand((H, W), generator=g)
    gt = torch.rand((3, H, W), generator=g)
    wvt = torch.eye(4)
    fovx, fovy = 0.9, 0.65
    fx = W / (2 * math.tan(fovx / 2.)); fy = H / (2 * math.tan(fovy / 2.))
    terms, dL = TB.train_loss(rend.contiguous(), gt, TB.window_taps(), wvt, float(fx), float(fy), *lambdas, True)
    ro = rend.clone().requires_grad_(True)
    out = O.training_loss(ro, gt, wvt, W, H, fovx, fovy, *lambdas)
    loss_o = out[0] if isinstance(out, (tuple, list)) else out
    (go,) = torch.autograd.grad(loss_o, ro)
    assert float(terms[0]) == pytest.approx(float(loss_o.detach()), rel=2e-5, abs=1e-7)
    _close(dL.numpy(), go.numpy(), 2e-4, "d loss / d rendering")


def test_emulated_adam_step_matches_oracle(emu):
    import train_epilogue._backend as TB
    rng = np.random.default_rng(3)
    for n in (1, 5, 1023, 4099):
        p = rng.normal(size=n).astype(np.float32); gr = rng.normal(size=n).astype(np.float32)
        m = (0.1 * rng.normal(size=n)).astype(np.float32); v = (0.01 * rng.uniform(size=n)).astype(np.float32)
        step, lr = 7, 0.016
        want_p, want_m, want_v = O.adam_step(p.copy(), gr, m.copy(), v.copy(), step, lr)
        tp, tg, tm, tv = (torch.from_numpy(a.copy()) for a in (p, gr, m, v))
        bc1 = 1 - 0.9 ** step; bc2 = 1 - 0.999 ** step
        TB.adam_step([(tp, tg, tm, tv, -(lr / bc1), bc2 ** 0.5)], 0.9, 0.999, 1e-15)      # (optim.py: step_size = -(lr / bias_correction1))
        _close(tp.numpy(), want_p, 2e-6, "param"); _close(tm.numpy(), want_m, 2e-6, "exp_avg"); _close(tv.numpy(), want_v, 2e-6, "exp_avg_sq")


def _cloud(n, seed, kind="uniform"):
    import test_knn
    return test_knn._cloud(n, seed, kind)


@pytest.mark.parametrize("n,kind", [(1, "uniform"), (2, "uniform"), (3, "uniform"), (4, "uniform"), (255, "uniform"), (257, "uniform"),
                                    (5000, "uniform"), (6000, "clustered"), (3000, "line"), (4000, "duplicates")])
def test_emulated_knn_matches_oracle(n, kind):
    lib = E.load()
    lib.gof_knn_ws_bytes.restype = C.c_size_t; lib.gof_knn_ws_bytes.argtypes = [C.c_int64]
    lib.gof_knn_mean_dist3.restype = C.c_int
    lib.gof_knn_mean_dist3.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    pts = np.ascontiguousarray(_cloud(n, n), np.float32)
    out = np.zeros(n, np.float32)
    nb = lib.gof_knn_ws_bytes(n)
    ws = E._aligned(nb)
    assert lib.gof_knn_mean_dist3(n, E._p(pts), E._p(out), E._p(ws), nb, None) == 0, lib.gof_last_error()
    want = KO.mean_dist3(pts)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(out), fin)
    np.testing.assert_allclose(out[fin], want[fin], rtol=2e-6)


def _mtets(lib, verts, tets, sdf, scales):
    V, Tt = len(verts), len(tets)
    t64 = np.ascontiguousarray(tets, np.int64); v32 = np.ascontiguousarray(verts, np.float32)
    s32 = np.ascontiguousarray(sdf, np.float32).reshape(-1); sc32 = np.ascontiguousarray(scales, np.float32).reshape(-1)
    tws = E._aligned(lib.gof_mtets_tet_ws_bytes(Tt))
    nv = C.c_int64(0)
    assert lib.gof_mtets_classify(V, Tt, E._p(t64), E._p(s32), E._p(tws), tws.size, C.byref(nv), None) == 0
    ews = E._aligned(lib.gof_mtets_edge_ws_bytes(int(nv.value)))
    ne, nf = C.c_int64(0), C.c_int64(0)
    assert lib.gof_mtets_count(V, Tt, E._p(t64), E._p(s32), E._p(tws), tws.size, E._p(ews), ews.size, C.byref(ne), C.byref(nf), None) == 0
    Ec, F = int(ne.value), int(nf.value)
    ids = np.zeros((Ec, 2), np.int64); pos = np.zeros((Ec, 2, 3), np.float32); esdf = np.zeros((Ec, 2, 1), np.float32)
    esc = np.zeros((Ec, 2, 1), np.float32); faces = np.zeros((F, 3), np.int64)
    p = lambda a: C.c_void_p(a.ctypes.data)   # noqa: E731
    if Ec or F:
        assert lib.gof_mtets_emit(V, Tt, E._p(t64), E._p(v32), E._p(s32), E._p(sc32), E._p(tws), tws.size, E._p(ews), ews.size, Ec, F,
                                  p(ids), p(pos), p(esdf), p(esc), p(faces), None) == 0
    return ids, pos, esdf, esc, faces


@pytest.mark.parametrize("n", [(3, 3, 3), (14, 12, 10)])
def test_emulated_marching_tets_match_oracle(n):
    lib = E.load()
    verts, tets = S.freudenthal_tets(*n)
    rng = np.random.default_rng(7)
    centre = np.array(n, np.float32) / 2
    sdf = (0.4 * min(n) - np.linalg.norm(verts - centre, axis=1) + rng.normal(0, 0.3, len(verts))).astype(np.float32)
    scales = rng.uniform(0.1, 1, len(verts)).astype(np.float32)
    got = _mtets(lib, verts, tets, sdf, scales)
    want = ob.marching_tets(verts, tets, sdf, scales)
    for a, b in zip(got, [want[0], want[1], want[2][..., None], want[3][..., None], want[4]]):
        assert np.array_equal(a, b)
    for const in (-1.0, 1.0):
        ids, pos, esdf, esc, faces = _mtets(lib, verts, tets, np.full(len(verts), const, np.float32), scales)
        assert len(ids) == 0 and len(faces) == 0


@pytest.fixture()
def emu_all(emu, monkeypatch):
    """... and the modules that bind the library handle at import (filter_3d, activations): handle swapped, the signatures the
    package declared on the product library mirrored onto the emulated one, torch.cuda.device() a no-op."""
    import importlib
    A = importlib.import_module("train_epilogue.activations")      # (the package re-exports functions of the same names)
    F = importlib.import_module("train_epilogue.filter_3d")
    A, F = sys.modules["train_epilogue.activations"], sys.modules["train_epilogue.filter_3d"]
    real = F.lib
    for mod in (F, A):
        monkeypatch.setattr(mod, "lib", emu)
    for name in ("gof_filter3d_ws_bytes", "gof_compute_3d_filter", "gof_add_densification_stats", "gof_act_scaling", "gof_act_scaling_backward",
                 "gof_act_opacity", "gof_act_opacity_backward", "gof_act_rotation", "gof_act_rotation_backward"):
        f_real, f_emu = getattr(real, name), getattr(emu, name)
        f_emu.argtypes, f_emu.restype = f_real.argtypes, f_real.restype
    monkeypatch.setattr(torch.cuda, "device", lambda *a, **k: contextlib.nullcontext())
    return emu


@pytest.mark.parametrize("P,ncam", [(1, 1), (1000, 3), (60_000, 12)])
def test_emulated_compute_3d_filter_matches_oracle(emu_all, P, ncam):
    import types
    import train_epilogue as T
    rng = np.random.default_rng(P + ncam)
    xyz = rng.uniform(-2.0, 2.0, (P, 3)).astype(np.float32)
    if P == 1:
        xyz[:] = 0.0
    cams = []
    for i in range(ncam):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        cams.append(types.SimpleNamespace(R=R, T=np.array([0.0, 0.0, 4.0]) + rng.normal(0, 0.2, 3), focal_x=float(rng.uniform(400, 1300)),
                                          focal_y=float(rng.uniform(400, 1300)), image_width=int(rng.integers(300, 1700)), image_height=int(rng.integers(300, 1100))))
    got = T.filter_3d(torch.from_numpy(xyz), T.camera_table(cams, "cpu")).numpy()
    ref = O.compute_3d_filter(torch.from_numpy(xyz), cams).numpy()
    got, ref = got.ravel(), ref.ravel()
    bad = np.abs(got - ref) > 2e-6 * np.abs(ref)           # (a point exactly on a threshold may flip for one camera: tests/test_train_epilogue_gpu.py)
    assert bad.mean() <= 1e-4, "%d of %d points differ" % (bad.sum(), bad.size)
    # nothing seen by any camera: fails like the reference (max of an empty tensor)
    cam = types.SimpleNamespace(R=np.eye(3), T=np.array([0.0, 0.0, -10.0]), focal_x=500.0, focal_y=500.0, image_width=640, image_height=480)
    with pytest.raises(RuntimeError):
        T.filter_3d(torch.zeros((10, 3)), T.camera_table([cam], "cpu"))


def test_emulated_add_densification_stats_matches_oracle(emu_all):
    import types
    import train_epilogue as T
    P = 50_003
    g = torch.Generator().manual_seed(12)
    st = [torch.zeros((P, 1)) for _ in range(4)]
    model = types.SimpleNamespace(xyz_gradient_accum=st[0].clone(), xyz_gradient_accum_abs=st[1].clone(), xyz_gradient_accum_abs_max=st[2].clone(), denom=st[3].clone())
    for it in range(4):
        grad = torch.randn((P, 3), generator=g) * (10.0 ** (-it * 3))
        grad[:, 2] = grad[:, 2].abs()
        filt = torch.rand(P, generator=g) < 0.6
        O.add_densification_stats(st[0], st[1], st[2], st[3], grad, filt)
        T.add_densification_stats(model, types.SimpleNamespace(grad=grad), filt)
    for name, ref in zip(("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom"), st):
        assert torch.allclose(getattr(model, name), ref, rtol=1e-6, atol=1e-30), name
    assert torch.equal(model.denom, st[3])


def test_emulated_activations_match_oracle(emu_all):
    import types
    import train_epilogue as T
    P = 20_000
    g = torch.Generator().manual_seed(44)
    rs = torch.randn((P, 3), generator=g) * 2.0 - 4.0
    ro = torch.randn((P, 1), generator=g) * 4.0
    rr = torch.randn((P, 4), generator=g); rr[:10] = 0.0
    f3 = torch.rand((P, 1), generator=g) * 0.1; f3[:1000] = 0.0
    ws, wo, wr = (torch.randn(s_, generator=g) for s_ in ((P, 3), (P, 1), (P, 4)))
    model = types.SimpleNamespace(_scaling=rs.clone().requires_grad_(True), _opacity=ro.clone().requires_grad_(True),
                                  _rotation=rr.clone().requires_grad_(True), filter_3D=f3)
    A = T.activations
    s, o, r = A.get_scaling_with_3D_filter(model), A.get_opacity_with_3D_filter(model), A.get_rotation(model)
    ((s * ws).sum() + (o * wo).sum() + (r * wr).sum()).backward()
    trs, tro, trr = (a.clone().requires_grad_(True) for a in (rs, ro, rr))
    so, oo, ro_ = O.scaling_with_3D_filter(trs, f3), O.opacity_with_3D_filter(tro, trs, f3), O.rotation(trr)
    gs, go, gr = torch.autograd.grad((so * ws).sum() + (oo * wo).sum() + (ro_ * wr).sum(), [trs, tro, trr])
    for a, ref, name in zip((s, o, r), (so, oo, ro_), ("scaling", "opacity", "rotation")):
        np.testing.assert_allclose(a.detach().numpy(), ref.detach().numpy(), rtol=3e-6, atol=1e-30, err_msg=name)
    for a, ref, name in zip((model._scaling.grad, model._opacity.grad, model._rotation.grad), (gs, go, gr), ("g_scaling", "g_opacity", "g_rotation")):
        assert np.isfinite(a.numpy()).all(), name
        _close(a.numpy(), ref.numpy(), 1e-5, name)


def test_emulated_densification_selection_and_row_surgery(emu_all, monkeypatch):
    """gof_densify_select / gof_compact_rows / gof_rows_gather run from source against the reference's masks restated with torch on
    the host (scene/gaussian_model.py:631-707): grads = accum / denom with NaN -> 0; selected = norm(grads) >= max_grad or
    norm(grads_abs) >= Q; clone if max(scale) <= size_threshold, else split; the three ordered index lists; rows rebuilt by one
    gather with new rows taken from `extra` (zeros for the Adam moments)."""
    D = sys.modules.get("train_epilogue.densify") or __import__("importlib").import_module("train_epilogue.densify")
    real = D.lib
    monkeypatch.setattr(D, "lib", emu_all)
    for name in ("gof_densify_ws_bytes", "gof_densify_select", "gof_compact_rows", "gof_rows_gather"):
        getattr(emu_all, name).argtypes, getattr(emu_all, name).restype = getattr(real, name).argtypes, getattr(real, name).restype
    g = torch.Generator().manual_seed(9)
    P = 20_011
    accum = torch.rand((P, 1), generator=g) * 3e-4
    accum_abs = torch.rand((P, 1), generator=g) * 6e-4
    denom = torch.randint(0, 5, (P, 1), generator=g).float()            # zeros: 0/0 = NaN -> 0
    scale_max = torch.rand(P, generator=g) * 0.2
    max_grad, size_threshold = 2e-4, 0.1
    grads = accum / denom; grads[grads.isnan()] = 0.0
    grads_abs = accum_abs / denom; grads_abs[grads_abs.isnan()] = 0.0
    ratio = (torch.norm(grads, dim=-1) >= max_grad).float().mean()
    Q = torch.quantile(grads_abs.reshape(-1), 1 - ratio)
    sel = (torch.norm(grads, dim=-1) >= max_grad) | (torch.norm(grads_abs, dim=-1) >= Q)
    clone = sel & (scale_max <= size_threshold)
    split = sel & (scale_max > size_threshold)
    role, keep_idx, clone_idx, split_idx = D.select(accum, accum_abs, denom, scale_max, max_grad, Q, size_threshold)
    assert torch.equal(role, clone.to(torch.uint8) + 2 * split.to(torch.uint8))
    assert torch.equal(clone_idx.long(), clone.nonzero().squeeze(1)) and torch.equal(split_idx.long(), split.nonzero().squeeze(1))
    assert torch.equal(keep_idx.long(), (~split).nonzero().squeeze(1))
    assert 0 < clone.sum() < P and 0 < split.sum() < P
    # compact_rows with and without an indirection; rows_gather with new rows
    keep = torch.rand(P, generator=g) < 0.7
    assert torch.equal(D.compact_rows(keep).long(), keep.nonzero().squeeze(1))
    src_rows = torch.randperm(P, generator=g).to(torch.int32)
    assert torch.equal(D.compact_rows(keep, src_rows), src_rows[keep])
    src = torch.randn((P, 15, 3), generator=g); extra = torch.randn((7, 15, 3), generator=g)
    rows = torch.cat([keep_idx[:1000], -(torch.arange(7, dtype=torch.int32) + 1), clone_idx[:50]])
    got = D.rows_gather(rows, src, extra)
    want = torch.cat([src[keep_idx[:1000].long()], extra, src[clone_idx[:50].long()]])
    assert torch.equal(got, want)
    got0 = D.rows_gather(rows, src, None)                                  # an Adam moment: zeros where new rows go
    want0 = want.clone(); want0[1000:1007] = 0
    assert torch.equal(got0, want0)
