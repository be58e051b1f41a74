"""CPU-only: the deferred loss of the launcher (train_epilogue/deferred.py) -- the inline loss of an UNCHANGED train.py (train.py:150-189)
evaluated by one fused call.  What is tested here is the LOGIC: which spellings stay deferred, that the coefficients of the script's
python arithmetic reach the fused call, and that every deviation from train.py's spelling computes eagerly with the right value and
gradient.  The four computing entry points (l1, ssim, depth_to_normal, the fused loss) are replaced by small differentiable torch
stand-ins -- the HIP ones have no CPU path; that the HIP fused loss equals the HIP eager composition is tests/test_train_epilogue_gpu.py's
and tests/test_e2e_scripts_gpu.py's subject.  The rasterizer is a stand-in autograd node that hands out the image exactly as
diff_gaussian_rasterization.rasterize_gaussians does (RenderedImage + its gradient slab)."""
import gc
import math
import types
import weakref

import pytest
import torch

import diff_gaussian_rasterization as DGR
from train_epilogue import deferred as D
from train_epilogue.fused_loss import TrainingLoss
from train_epilogue.pose import PoseMatrix

H, W = 12, 16


class _Raster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.slab = DGR._GradSlab()
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        return ctx.slab.deliver(g)


def render(x):
    color = _Raster.apply(x)
    fn = color.grad_fn
    return DGR._as_rendered_image(color, getattr(fn, "slab", None) if fn is not None else None)        # (as rasterize_gaussians does)


# ---- stand-ins for the computing entry points (any differentiable functions do: eager and fused must only be the SAME functions)
def _l1(a, b):
    return torch.abs(a - b).mean()


def _ssim(a, b, window_size=11, size_average=True):
    return 1.0 - ((a - b) ** 2).mean() * 0.5


def _depth_to_normal(view, depth):
    d = depth.reshape(H, W)
    n = torch.stack([d, 2.0 * d + 0.1, 1.0 + d * d], dim=-1)
    return n, torch.zeros_like(n)


def _composition(rendering, gt, view, l_dssim, l_dn, l_dist, normalize_p=2):
    """train.py:150-188 on plain tensors with the stand-ins"""
    image = rendering[:3]
    Ll1, s = _l1(image, gt), _ssim(image, gt)
    rgb = (1.0 - l_dssim) * Ll1 + l_dssim * (1.0 - s)
    dist = rendering[8].mean()
    dn = _depth_to_normal(view, rendering[6][None])[0].permute(2, 0, 1)
    unit = torch.nn.functional.normalize(rendering[3:6], p=normalize_p, dim=0)
    c2w = torch.linalg.inv(torch.Tensor(view.world_view_transform.as_subclass(torch.Tensor)).T)
    world = (c2w[:3, :3] @ unit.reshape(3, -1)).reshape(3, H, W)
    dnl = (1 - (world * dn).sum(dim=0)).mean()
    return rgb + dnl * l_dn + dist * l_dist, Ll1, s, rgb, dnl, dist


calls = {"fused": 0}


def _fused(rendering, gt, view, l_dssim, l_dn, l_dist):
    calls["fused"] += 1
    calls["lambdas"] = (l_dssim, l_dn, l_dist)
    r = rendering.detach().clone().requires_grad_(True)
    terms = _composition(r, gt, view, l_dssim, l_dn, l_dist)
    (dL,) = torch.autograd.grad(terms[0], r)
    return TrainingLoss(*(t.detach() for t in terms)), dL


@pytest.fixture(autouse=True)
def deferred_on():
    keep = dict(D.impl)
    D.impl.update({"l1": _l1, "ssim": _ssim, "depth_to_normal": _depth_to_normal, "fused": _fused})
    D.enable(True)
    for k in D.stats:
        D.stats[k] = 0
    calls["fused"] = 0
    yield
    D.enable(False)
    D.impl.update(keep)


def _setup(seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand((9, H, W), generator=g) + 0.1).requires_grad_(True)
    gt = torch.rand((3, H, W), generator=g)
    a = 0.3
    R = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
    wvt = torch.eye(4)
    wvt[:3, :3] = R
    wvt[3, :3] = torch.tensor([0.1, -0.2, 0.5])
    view = types.SimpleNamespace(world_view_transform=PoseMatrix.wrap(wvt), image_width=W, image_height=H, FoVx=0.8, FoVy=0.6)
    return x, gt, view


def _expected(x, gt, view, *lambdas, **kw):
    xr = x.detach().clone().requires_grad_(True)
    terms = _composition(xr, gt, view, *lambdas, **kw)
    terms[0].backward()
    return [float(t.detach()) for t in terms], xr.grad


def _train_py_lines(rendering, target, cam, lambda_dssim, lambda_depth_normal, lambda_distortion, l1_loss=D.l1_loss, ssim=D.ssim,
                    depth_to_normal=D.depth_to_normal):
    """the statements of train.py:151-188 as the script spells them (the helpers are what the launcher binds)"""
    img = rendering[:3, :, :]
    l1_term = l1_loss(img, target)
    rgb_term = (1.0 - lambda_dssim) * l1_term + lambda_dssim * (1.0 - ssim(img, target))
    dmap = rendering[8, :, :]
    dist_term = dmap.mean()
    zmap = rendering[6, :, :]
    n_from_depth, _ = depth_to_normal(cam, zmap[None, ...])
    n_from_depth = n_from_depth.permute(2, 0, 1)
    n_img = rendering[3:6, :, :]
    n_img = torch.nn.functional.normalize(n_img, p=2, dim=0)
    pose_inv = (cam.world_view_transform.T).inverse()
    n_flat = pose_inv[:3, :3] @ n_img.reshape(3, -1)
    n_world = n_flat.reshape(3, *n_img.shape[1:])
    n_err = 1 - (n_world * n_from_depth).sum(dim=0)
    dn_term = n_err.mean()
    loss = rgb_term + dn_term * lambda_depth_normal + dist_term * lambda_distortion
    return loss, l1_term


@pytest.mark.parametrize("lambdas", [(0.2, 0.05, 100.0), (0.2, 0.0, 0.0), (0.35, 0.05, 0.0)])
def test_the_scripts_own_lines_become_one_fused_call(lambdas):
    x, gt, view = _setup()
    loss, Ll1 = _train_py_lines(render(x), gt, view, *lambdas)
    assert isinstance(loss, D.DeferredLoss) and isinstance(Ll1, D.DeferredLoss)
    assert D.stats == {"fused_backwards": 0, "eager_terms": 0, "eager_tensors": 0} and calls["fused"] == 0      # nothing computed so far
    loss.backward()
    assert calls["fused"] == 1 and D.stats == {"fused_backwards": 1, "eager_terms": 0, "eager_tensors": 0}
    assert calls["lambdas"] == pytest.approx(lambdas, rel=1e-12)
    want, grad = _expected(x, gt, view, *lambdas)
    assert torch.allclose(x.grad, grad, rtol=1e-6, atol=1e-9)
    assert loss.item() == pytest.approx(want[0], rel=1e-5) and Ll1.item() == pytest.approx(want[1], rel=1e-6)      # train.py:239, :305
    assert float(loss.detach()) == pytest.approx(want[0], rel=1e-5)
    with pytest.raises(RuntimeError, match="differentiated already"):
        loss.eager()


def test_a_scaled_combination_scales_the_gradient():
    """coefficients with a_l1 - a_ssim != 1: the fused call gets the normalised lambdas, the gradient the scale"""
    x, gt, view = _setup(1)
    loss, _ = _train_py_lines(render(x), gt, view, 0.2, 0.05, 10.0)
    loss = 3.0 * loss / 2 - 0.25
    loss.backward()
    assert D.stats["fused_backwards"] == 1 and calls["lambdas"] == pytest.approx((0.2, 0.05, 10.0), rel=1e-12)
    want, grad = _expected(x, gt, view, 0.2, 0.05, 10.0)
    assert torch.allclose(x.grad, 1.5 * grad, rtol=1e-6, atol=1e-9)
    assert loss.item() == pytest.approx(1.5 * want[0] - 0.25, rel=1e-5)


def _deviations():
    def other_gt(r, gt, view):
        return _train_py_lines(r, gt, view, 0.2, 0.05, 100.0, ssim=lambda a, b: D.ssim(a, b.clone()))[0]

    def tensor_factor(r, gt, view):
        return _train_py_lines(r, gt, view, 0.2, 0.05, 100.0)[0] * torch.tensor(1.0)

    def extra_term(r, gt, view):
        return _train_py_lines(r, gt, view, 0.2, 0.05, 100.0)[0] + 0.0 * (r[7] ** 2).mean()

    def value_first(r, gt, view):
        loss = _train_py_lines(r, gt, view, 0.2, 0.05, 100.0)[0]
        assert math.isfinite(loss.item())            # a read-back BEFORE the backward: the eager composition
        return loss

    def plain_pose(r, gt, view):
        """GOF_PLAIN_POSE=1: the camera's pose is a plain tensor, `.T.inverse()` a fresh matrix nobody can vouch for"""
        plain = types.SimpleNamespace(**dict(vars(view), world_view_transform=view.world_view_transform.as_subclass(torch.Tensor).clone()))
        return _train_py_lines(r, gt, plain, 0.2, 0.05, 100.0)[0]

    def tensor_valued_ssim(r, gt, view):
        return _train_py_lines(r, gt, view, 0.2, 0.05, 100.0, ssim=lambda a, b: D.ssim(a, b) * torch.ones(()))[0]
    return {"other_gt": other_gt, "tensor_factor": tensor_factor, "extra_term": extra_term, "value_first": value_first, "tensor_valued_ssim": tensor_valued_ssim,
            "plain_pose": plain_pose}


PARTLY_FUSED = ("extra_term", "other_gt", "plain_pose")       # a 0-dim tensor joins the combination: carried along, the rest is still one fused call


@pytest.mark.parametrize("name", sorted(_deviations()))
def test_a_different_spelling_computes_eagerly_with_the_same_result(name):
    x, gt, view = _setup(2)
    loss = _deviations()[name](render(x), gt, view)
    loss.backward()
    if name in PARTLY_FUSED:
        assert isinstance(loss, D.DeferredLoss) and loss.extra is not None and calls["fused"] == 1 and D.stats["fused_backwards"] == 1
        assert D.stats["eager_tensors"] >= (1 if name == "plain_pose" else 0)        # (the normal chain computed by torch; other_gt's ssim by the helper itself)
    else:
        assert calls["fused"] == 0 and D.stats["fused_backwards"] == 0 and D.stats["eager_terms"] >= 1
    want, grad = _expected(x, gt, view, 0.2, 0.05, 100.0)
    assert torch.allclose(x.grad, grad, rtol=1e-5, atol=1e-8)
    assert float(loss) == pytest.approx(want[0], rel=1e-5)


def test_the_decoupled_appearance_l1_is_carried_along_and_the_rest_stays_fused():
    """train.py:156-161 with dataset.use_decoupled_appearance (the reference's own TNT / DTU runs): the deferred L1 is REPLACED by one that goes
    through a network (here: a small conv on a crop of the image) and stays torch code; the sum `0.8 * tensor + 0.2 * (1 - ssim)` keeps the
    tensor as a carried term, `loss.backward()` is one fused call for ssim / normal consistency / distortion (with lambda_dssim = 1 and the
    scale 0.2: no L1 inside it) plus the tensor's own graph, in ONE pass -- the network's parameters get their gradients, the image its sum."""
    x, gt, view = _setup(7)
    net = torch.nn.Conv2d(3, 3, 3, padding=1)
    torch.manual_seed(0)
    for p_ in net.parameters():
        torch.nn.init.uniform_(p_, -0.3, 0.3)

    def appearance_l1(img, target):
        crop, tcrop = img[:, 2:10, 3:11], target[:, 2:10, 3:11]
        return _l1(torch.sigmoid(net(crop[None]))[0] * crop, tcrop)

    def run(rendering, l1_loss, ssim, depth_to_normal):
        img = rendering[:3, :, :]
        l1_term = l1_loss(img, gt)                    # (train.py:156: evaluated, then overwritten)
        l1_term = appearance_l1(img, gt)              # (train.py:158-159)
        rgb_term = (1.0 - 0.2) * l1_term + 0.2 * (1.0 - ssim(img, gt))
        dist_term = rendering[8, :, :].mean()
        zmap = rendering[6, :, :]
        n_from_depth = depth_to_normal(view, zmap[None, ...])[0].permute(2, 0, 1)
        n_img = torch.nn.functional.normalize(rendering[3:6, :, :], p=2, dim=0)
        pose_inv = (view.world_view_transform.T).inverse()
        n_world = (pose_inv[:3, :3] @ n_img.reshape(3, -1)).reshape(3, *n_img.shape[1:])
        dn_term = (1 - (n_world * n_from_depth).sum(dim=0)).mean()
        return rgb_term + dn_term * 0.05 + dist_term * 100.0

    loss = run(render(x), D.l1_loss, D.ssim, D.depth_to_normal)
    assert isinstance(loss, D.DeferredLoss) and loss.extra is not None and "l1" not in {t for t, a in loss.coef.items() if a != 0}
    loss.backward()
    assert calls["fused"] == 1 and D.stats["eager_terms"] == 0 and calls["lambdas"] == pytest.approx((1.0, 0.05 / 0.2, 100.0 / 0.2), rel=1e-9)
    got_x, got_w, value = x.grad.clone(), net.weight.grad.clone(), loss.item()
    x.grad = None
    net.weight.grad = None
    D.enable(False)
    ref = run(render(x), _l1, _ssim, _depth_to_normal)
    ref.backward()
    assert torch.allclose(got_x, x.grad, rtol=1e-5, atol=1e-8) and torch.allclose(got_w, net.weight.grad, rtol=1e-5, atol=1e-9)
    assert value == pytest.approx(ref.item(), rel=1e-5)


def test_the_normal_chain_spelled_differently_is_eager():
    """p=1 normalisation, a matrix that is not the camera's inverse pose, a sum over another dimension: the intermediate materialises and
    the script goes on with plain tensors"""
    x, gt, view = _setup(3)
    r = render(x)
    unit = torch.nn.functional.normalize(r[3:6, :, :], p=1, dim=0)
    assert type(unit) is torch.Tensor
    unit = torch.nn.functional.normalize(r[3:6, :, :], p=2, dim=0)
    assert isinstance(unit, D.DeferredTensor) and unit.shape == (3, H, W)
    dn, _ = D.depth_to_normal(view, r[6, :, :][None, ...])
    assert isinstance(dn, D.DeferredTensor) and dn.shape == (H, W, 3)
    c2w = (view.world_view_transform.T).inverse()
    other = torch.eye(3)
    flat = unit.reshape(3, -1)
    assert type(other @ flat) is torch.Tensor                       # a plain matrix: computed
    world = (c2w[:3, :3] @ flat).reshape(3, H, W)
    assert isinstance(world, D.DeferredTensor) and world.kind == "world"
    prod = world * dn.permute(2, 0, 1)
    assert isinstance(prod, D.DeferredTensor)
    assert type(prod.sum(dim=1)) is torch.Tensor
    err = 1 - prod.sum(dim=0)
    assert isinstance(err, D.DeferredTensor) and type(2 - prod.sum(dim=0)) is torch.Tensor
    loss = err.mean()
    assert isinstance(loss, D.DeferredLoss)
    (loss * 1.0).backward()                  # no l1 / ssim term: nothing the fused call could be given -> eager
    assert calls["fused"] == 0
    xr = x.detach().clone().requires_grad_(True)
    un = torch.nn.functional.normalize(xr[3:6], p=2, dim=0)
    cw = torch.linalg.inv(view.world_view_transform.as_subclass(torch.Tensor).T)
    e = (1 - ((cw[:3, :3] @ un.reshape(3, -1)).reshape(3, H, W) * _depth_to_normal(view, xr[6][None])[0].permute(2, 0, 1)).sum(dim=0)).mean()
    e.backward()
    assert torch.allclose(x.grad, xr.grad, rtol=1e-5, atol=1e-8)


def test_without_gradients_and_when_disabled_everything_is_plain():
    x, gt, view = _setup(4)
    with torch.no_grad():
        r = render(x)
        assert type(D.l1_loss(r[:3], gt)) is torch.Tensor and type(r[8].mean()) is torch.Tensor
    D.enable(False)
    r = render(x)
    loss, _ = _train_py_lines(r, gt, view, 0.2, 0.05, 100.0)
    assert type(loss) is torch.Tensor and type(r[:3]) is torch.Tensor
    loss.backward()
    want, grad = _expected(x, gt, view, 0.2, 0.05, 100.0)
    assert torch.allclose(x.grad, grad, rtol=1e-5, atol=1e-8) and calls["fused"] == 0


def test_channel_slices_behave_as_plain_tensors_otherwise():
    x, gt, view = _setup(5)
    r = render(x)
    img = r[:3, :, :]
    assert isinstance(img, D.RenderedChannels)
    for out in (img + 1, img.clamp(0, 1), torch.cat([img, img], dim=2), img[:, 1:3], img.detach(), r[7, :, :, None], r[6].mean(), r[8].mean(dim=0)):
        assert type(out) is torch.Tensor
    (img.sum() + r[7].sum() * 2).backward()
    want = torch.zeros(9, H, W)
    want[:3] = 1
    want[7] = 2
    assert torch.equal(x.grad, want)


def test_a_frame_is_released_by_reference_counting_after_its_backward():
    x, gt, view = _setup(6)
    gc.collect()
    gc.disable()
    try:
        r = render(x)
        loss, Ll1 = _train_py_lines(r, gt, view, 0.2, 0.05, 100.0)
        loss.backward()
        ref = weakref.ref(r)
        value = loss.item()
        del r, loss, Ll1
        assert ref() is None, "the image is still referenced after its loss went out of scope (a reference cycle)"
        assert math.isfinite(value)
    finally:
        gc.enable()
