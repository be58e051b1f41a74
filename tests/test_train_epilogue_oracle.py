"""CPU-only: the training-epilogue oracle (oracle/train_epilogue_oracle.py) against golden vectors produced by
EXECUTING the reference's own utils/loss_utils.py, utils/depth_utils.py and torch.optim.Adam
(tests/golden/make_golden_train.py), plus the host-side behaviour of the product mirrors (argument errors)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import train_epilogue_oracle as O   # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "ref_train_epilogue_golden.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_ssim_and_l1_oracle_match_reference_python(tag):
    x = torch.from_numpy(G[f"ssim_{tag}_x"]).requires_grad_(True)
    y = torch.from_numpy(G[f"ssim_{tag}_y"])
    s = O.ssim(x, y)
    (gx,) = torch.autograd.grad(s, x)
    assert s.item() == pytest.approx(float(G[f"ssim_{tag}_value"]), rel=1e-6)
    np.testing.assert_allclose(gx.numpy(), G[f"ssim_{tag}_grad"], rtol=1e-5, atol=1e-9)
    assert O.l1_loss(x.detach(), y).item() == pytest.approx(float(G[f"l1_{tag}_value"]), rel=1e-6)


def test_ssim_batched_oracle_matches_reference_python():
    x = torch.from_numpy(G["ssim_batch_x"]).requires_grad_(True)
    y = torch.from_numpy(G["ssim_batch_y"])
    s = O.ssim(x, y, size_average=False)
    (gx,) = torch.autograd.grad((s * torch.from_numpy(G["ssim_batch_w"])).sum(), x)
    np.testing.assert_allclose(s.detach().numpy(), G["ssim_batch_value"], rtol=1e-6)
    np.testing.assert_allclose(gx.numpy(), G["ssim_batch_grad"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_depth_to_normal_oracle_matches_reference_python(tag):
    W, H, fovx, fovy = G[f"dn_{tag}_cam"]
    W, H = int(W), int(H)
    wvt = torch.from_numpy(G[f"dn_{tag}_wvt"])
    depth = torch.from_numpy(G[f"dn_{tag}_depth"]).requires_grad_(True)
    normals, points = O.depth_to_normal(wvt, W, H, fovx, fovy, depth)
    f = (normals * torch.from_numpy(G[f"dn_{tag}_wn"])).sum() + (points * torch.from_numpy(G[f"dn_{tag}_wp"])).sum()
    (gd,) = torch.autograd.grad(f, depth)
    np.testing.assert_array_equal(normals.detach().numpy(), G[f"dn_{tag}_normals"])
    np.testing.assert_array_equal(points.detach().numpy(), G[f"dn_{tag}_points"])
    np.testing.assert_array_equal(gd.numpy(), G[f"dn_{tag}_grad"])
    np.testing.assert_array_equal(O.depths_to_points(wvt, W, H, fovx, fovy, depth.detach()).numpy(), G[f"dn_{tag}_points_flat"])


GL = np.load(os.path.join(ROOT, "tests", "golden", "ref_train_loss_golden.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_training_loss_oracle_matches_reference_python(tag):
    """train.py:150-188 restated (oracle) vs the same statements executed over the reference's own l1_loss / ssim / depth_to_normal."""
    W, H, fovx, fovy = GL[f"{tag}_cam"]
    l_dssim, l_dn, l_dist = GL[f"{tag}_lambdas"]
    r = torch.from_numpy(GL[f"{tag}_rendering"]).requires_grad_(True)
    terms = O.training_loss(r, torch.from_numpy(GL[f"{tag}_gt"]), torch.from_numpy(GL[f"{tag}_wvt"]), int(W), int(H), fovx, fovy,
                            float(l_dssim), float(l_dn), float(l_dist))
    terms[0].backward()
    np.testing.assert_allclose([t.item() for t in terms], GL[f"{tag}_terms"], rtol=1e-6)
    np.testing.assert_allclose(r.grad.numpy(), GL[f"{tag}_grad"], rtol=1e-5, atol=1e-10)


def test_adam_oracle_matches_torch_adam():
    """fp32 numpy restatement vs torch.optim.Adam (single-tensor CPU implementation) over 3 steps, per-group lr."""
    sizes, lrs = G["adam_sizes"], G["adam_lrs"]
    p = G["adam_p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    off = np.concatenate([[0], np.cumsum(sizes)])
    for step in range(3):
        g = G[f"adam_g{step}"]
        for i, lr in enumerate(lrs):
            sl = slice(off[i], off[i + 1])
            p[sl], m[sl], v[sl] = O.adam_step(p[sl], g[sl], m[sl], v[sl], step + 1, lr)
        np.testing.assert_allclose(p, G[f"adam_p{step + 1}"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(m, G["adam_m3"], rtol=2e-6, atol=1e-8)      # lerp cancels when grad ~ -exp_avg
    np.testing.assert_allclose(v, G["adam_v3"], rtol=2e-6, atol=1e-20)


def _cams_from_table(tab):
    return [types.SimpleNamespace(R=r[0:9].reshape(3, 3), T=r[9:12], focal_x=float(r[12]), focal_y=float(r[13]),
                                  image_width=int(r[14]), image_height=int(r[15])) for r in np.asarray(tab, dtype=np.float64)]


def test_compute_3d_filter_oracle_matches_reference_python():
    """GaussianModel.compute_3D_filter executed by the reference itself (golden) vs the restatement: same torch ops -> bit-equal."""
    got = O.compute_3d_filter(torch.from_numpy(G["f3d_xyz"]), _cams_from_table(G["f3d_cams"]))
    assert got.shape == G["f3d_filter"].shape
    np.testing.assert_array_equal(got.numpy(), G["f3d_filter"])
    assert len(np.unique(G["f3d_filter"])) > 1000           # a non-trivial field, and unseen points exist in the fixture:
    assert (G["f3d_filter"] == G["f3d_filter"].max()).sum() > 1


def test_activation_oracle_matches_reference_python():
    """get_scaling_with_3D_filter / get_opacity_with_3D_filter / get_rotation: the reference's own property bodies (golden)."""
    rs = torch.from_numpy(G["act_raw_scaling"]).requires_grad_(True)
    ro = torch.from_numpy(G["act_raw_opacity"]).requires_grad_(True)
    rr = torch.from_numpy(G["act_raw_rotation"]).requires_grad_(True)
    f3 = torch.from_numpy(G["act_filter_3D"])
    s, o, r = O.scaling_with_3D_filter(rs, f3), O.opacity_with_3D_filter(ro, rs, f3), O.rotation(rr)
    np.testing.assert_array_equal(s.detach().numpy(), G["act_scaling"])
    np.testing.assert_array_equal(o.detach().numpy(), G["act_opacity"])
    np.testing.assert_array_equal(r.detach().numpy(), G["act_rotation"])
    f = (s * torch.from_numpy(G["act_w_s"])).sum() + (o * torch.from_numpy(G["act_w_o"])).sum() + (r * torch.from_numpy(G["act_w_r"])).sum()
    gs, go, gr = torch.autograd.grad(f, [rs, ro, rr])
    np.testing.assert_array_equal(gs.numpy(), G["act_g_scaling"])
    np.testing.assert_array_equal(go.numpy(), G["act_g_opacity"])
    np.testing.assert_array_equal(gr.numpy(), G["act_g_rotation"])


# ---- host-side behaviour of the product mirrors (no kernel is launched) ---------------------------------
def test_mirrors_fail_loudly_without_a_device_and_on_bad_arguments():
    import train_epilogue as T
    x = torch.rand(3, 20, 20)
    with pytest.raises(RuntimeError, match="ROCm device"):
        T.ssim(x, x)
    with pytest.raises(NotImplementedError):
        T.ssim(x, x, window_size=7)
    view = types.SimpleNamespace(world_view_transform=torch.eye(4), image_width=20, image_height=20, FoVx=0.8, FoVy=0.8)
    with pytest.raises(RuntimeError, match="ROCm device"):
        T.depth_to_normal(view, torch.rand(1, 20, 20))
    with pytest.raises(RuntimeError, match="invalid"):
        T.depth_to_normal(view, torch.rand(1, 10, 10))
    p = torch.nn.Parameter(torch.rand(5))
    with pytest.raises(ValueError, match="Invalid epsilon"):
        T.FusedAdam([p], eps=-1.0)
    with pytest.raises(NotImplementedError):
        T.FusedAdam([p], weight_decay=0.1)
    opt = T.FusedAdam([{"params": [p], "lr": 0.1, "name": "xyz"}], lr=0.0, eps=1e-15)     # gaussian_model.py:349-360
    assert opt.param_groups[0]["name"] == "xyz" and opt.param_groups[0]["eps"] == 1e-15
    opt.step()                      # no grad -> nothing to do, as torch
    p.grad = torch.ones(5)
    with pytest.raises(RuntimeError, match="ROCm device"):
        opt.step()
    assert torch.optim.Adam([p], lr=0.0, eps=1e-15).state_dict()["param_groups"][0].keys() <= opt.state_dict()["param_groups"][0].keys() | {"decoupled_weight_decay"}
