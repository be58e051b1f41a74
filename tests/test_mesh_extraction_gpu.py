"""The view loop of the mesh extraction with its reduction fused into the point pass (reference extract_mesh.py:17-34,
evaluage_alpha): `integrate_min_into` / `mesh_extraction.evaluate_alpha` against the reference's own
composition -- per-view outputs combined with torch.min / torch.where -- bit for bit."""
import math
import numpy as np
import pytest
import torch

from gpu_common import *   # noqa: F401,F403

pytestmark = pytest.mark.gpu


def _views(sc, n):
    """n cameras looking at the frustum scene: the original one and copies translated / rotated a little, so that every view sees
    most points, some points leave some views, and the minimum is attained in different views."""
    import synthetic_scenes as S
    W, H = sc["W"], sc["H"]
    fovx, fovy = 2 * math.atan(sc["tanfovx"]), 2 * math.atan(sc["tanfovy"])
    out = []
    for k in range(n):
        ang = 0.06 * k
        R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
        T = np.array([0.15 * k, -0.1 * k, 0.2 * k])
        cam = S.camera(W, H, fovx, fovy, R=R, T=T)
        v = dict(sc)
        v.update(cam)
        out.append(v)
    return out


def _rasterizers(sc, n):
    from diff_gaussian_rasterization import GaussianRasterizer
    sds = [to_dev(v) for v in _views(sc, n)]
    return sds, [GaussianRasterizer(settings_from(sd)) for sd in sds]


def _call(r, sd, pts):
    return r.integrate(points3D=pts, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"],
                       scales=sd["scales"], rotations=sd["rotations"])


@pytest.mark.parametrize("with_color", [False, True])
@pytest.mark.parametrize("cached", [False, True])
def test_min_over_views_fused_into_the_point_pass_is_bit_identical(with_color, cached):
    import synthetic_scenes as S
    import diff_gaussian_rasterization as DGR
    sc = S.scene_frustum(60_000, W=640, H=416, focal=480.0, seed=3)
    sds, rs = _rasterizers(sc, 4)
    allpts = S.tetra_points(sc)
    rng = np.random.default_rng(0)
    pts = torch.from_numpy(np.ascontiguousarray(allpts[rng.choice(len(allpts), 200_000, replace=False)])).cuda()
    N = pts.shape[0]
    # the reference's composition (extract_mesh.py:18-31)
    final_alpha = torch.ones(N, device="cuda")
    final_color = torch.ones(N, 3, device="cuda")
    improved = []
    for r, sd in zip(rs, sds):
        _, alpha_i, color_i, _ = _call(r, sd, pts)
        improved.append(int((alpha_i < final_alpha).sum()))
        final_color = torch.where((alpha_i < final_alpha).reshape(-1, 1), color_i, final_color)
        final_alpha = torch.min(final_alpha, alpha_i)
    assert (final_alpha < 1).float().mean() > 0.5 and sum(1 for n in improved if n > 1000) >= 3, improved
    # fused
    acc_alpha = torch.ones(N, device="cuda")
    acc_color = torch.ones(N, 3, device="cuda") if with_color else None
    DGR.integrate_view_cache().clear()
    for rounds in range(2 if cached else 1):          # second round: every view is served from the per-view cache
        acc_alpha.fill_(1.0)
        if acc_color is not None:
            acc_color.fill_(1.0)
        with DGR.integrate_min_into(acc_alpha, acc_color):
            for k, (r, sd) in enumerate(zip(rs, sds)):
                if cached:
                    with DGR.integrate_view_key(("mesh-test", k)):
                        ret = _call(r, sd, pts)
                else:
                    ret = _call(r, sd, pts)
                assert ret[1] is acc_alpha and ret[2] is acc_color
        assert torch.equal(acc_alpha, final_alpha)
        if with_color:
            assert torch.equal(acc_color, final_color)
    if cached:
        assert DGR.integrate_view_cache().hits >= 4
    DGR.integrate_view_cache().clear()
    # outside the context the call is the plain one again
    _, alpha_plain, _, _ = _call(rs[0], sds[0], pts)
    assert alpha_plain is not acc_alpha


def test_nan_and_points_outside_a_view_follow_torch_semantics():
    """A point outside a view keeps its running values; a NaN alpha (degenerate Gaussian) poisons the minimum as torch.min does and
    leaves the colour alone (`alpha < final_alpha` is false)."""
    import synthetic_scenes as S
    import diff_gaussian_rasterization as DGR
    sc = S.scene_frustum(5_000, W=320, H=208, focal=240.0, seed=5)
    sds, rs = _rasterizers(sc, 2)
    pts_np = S.tetra_points(sc)[:20_000].copy()
    pts_np[:100] += np.array([1e4, 0, 0], dtype=np.float32)         # far outside every view
    pts = torch.from_numpy(np.ascontiguousarray(pts_np)).cuda()
    N = pts.shape[0]
    acc_alpha = torch.ones(N, device="cuda"); acc_color = torch.full((N, 3), 0.25, device="cuda")
    with DGR.integrate_min_into(acc_alpha, acc_color):
        for r, sd in zip(rs, sds):
            _call(r, sd, pts)
    assert torch.all(acc_alpha[:100] == 1) and torch.all(acc_color[:100] == 0.25)
    fa = torch.ones(N, device="cuda"); fc = torch.full((N, 3), 0.25, device="cuda")
    for r, sd in zip(rs, sds):
        _, a, c, _ = _call(r, sd, pts)
        fc = torch.where((a < fa).reshape(-1, 1), c, fc); fa = torch.min(fa, a)
    assert torch.equal(acc_alpha, fa) and torch.equal(acc_color, fc)
    # wrong buffers are refused
    with pytest.raises(RuntimeError):
        with DGR.integrate_min_into(torch.ones(N + 1, device="cuda")):
            _call(rs[0], sds[0], pts)


def test_evaluate_alpha_matches_the_scripts_function():
    """mesh_extraction.evaluate_alpha == extract_mesh.py:17-34 with the same `integrate` callable."""
    import synthetic_scenes as S
    import mesh_extraction
    sc = S.scene_frustum(20_000, W=480, H=320, focal=360.0, seed=8)
    sds, rs = _rasterizers(sc, 3)
    pts = torch.from_numpy(np.ascontiguousarray(S.tetra_points(sc)[:90_000])).cuda()
    views = list(range(3))

    def integrate(points, view, gaussians, pipeline, background, kernel_size):
        color, alpha_i, color_i, radii = _call(rs[view], sds[view], points)
        return {"render": color, "alpha_integrated": alpha_i, "color_integrated": color_i, "radii": radii}

    def reference(points, return_color):                     # extract_mesh.py:17-34, verbatim semantics
        final_alpha = torch.ones((points.shape[0]), dtype=torch.float32, device="cuda")
        final_color = torch.ones((points.shape[0], 3), dtype=torch.float32, device="cuda")
        for view in views:
            ret = integrate(points, view, None, None, None, kernel_size=0.0)
            if return_color:
                final_color = torch.where((ret["alpha_integrated"] < final_alpha).reshape(-1, 1), ret["color_integrated"], final_color)
            final_alpha = torch.min(final_alpha, ret["alpha_integrated"])
        return (1 - final_alpha, final_color) if return_color else 1 - final_alpha

    a = mesh_extraction.evaluate_alpha(pts, views, None, None, None, 0.0, integrate=integrate, progress=False)
    assert torch.equal(a, reference(pts, False))
    a2, c2 = mesh_extraction.evaluate_alpha(pts, views, None, None, None, 0.0, return_color=True, integrate=integrate, progress=False)
    ra, rc = reference(pts, True)
    assert torch.equal(a2, ra) and torch.equal(c2, rc)
