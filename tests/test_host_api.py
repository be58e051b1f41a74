"""CPU-only: the Python surface mirrors the reference's operator API (names, argument meaning,
error behaviour) -- reference diff_gaussian_rasterization/__init__.py:167-305."""
import inspect

import pytest
import torch

import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def _settings(H=8, W=8):
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.5, kernel_size=0.0,
        subpixel_offset=torch.zeros(H, W, 2), bg=torch.zeros(3), scale_modifier=1.0,
        viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
        prefiltered=False, debug=False)


def test_settings_fields_and_order():
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "kernel_size", "subpixel_offset", "bg",
        "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")


def test_rasterizer_signatures():
    fwd = list(inspect.signature(GaussianRasterizer.forward).parameters)
    assert fwd == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations",
                   "cov3D_precomp", "view2gaussian_precomp"]
    integ = list(inspect.signature(GaussianRasterizer.integrate).parameters)
    assert integ == ["self", "points3D"] + fwd[1:]
    assert hasattr(GaussianRasterizer, "markVisible")
    assert list(inspect.signature(dgr.rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp",
        "view2gaussian_precomp", "raster_settings"]


def test_exactly_one_colour_source():
    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), shs=torch.zeros(4, 1, 3), colors_precomp=x, scales=x,
          rotations=torch.zeros(4, 4))


def test_exactly_one_covariance_source():
    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4),
          cov3D_precomp=torch.zeros(4, 6))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r.integrate(points3D=x, means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x, scales=x)


def test_means3D_shape_error_matches_reference_message():
    r = GaussianRasterizer(_settings())
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(means3D=torch.zeros(4, 2), means2D=torch.zeros(4, 3), opacities=torch.zeros(4, 1), colors_precomp=torch.zeros(4, 3),
          scales=torch.zeros(4, 3), rotations=torch.zeros(4, 4))


def test_no_cpu_fallback():
    """There is no CPU render path (as in the reference, gaussian_renderer/__init__.py:26,37): host tensors are rejected."""
    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="ROCm device"):
        r(means3D=x, means2D=x, opacities=torch.zeros(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4))


def test_product_does_not_import_oracle():
    """The oracle and the host emulation of the kernels (tests/hipemu) are test infrastructure: nothing under
    gaussian-opacity-fields_amd/ may reference either."""
    import os
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(dgr.__file__)))
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "oracle_binding" not in text and "gofref_" not in text, os.path.join(dirpath, f)
                assert "hipemu" not in text.lower() and "libgof_hip_emu" not in text, os.path.join(dirpath, f)


def test_num_rendered_is_the_count_and_carries_the_layout_size():
    """RasterizeGaussiansCUDA returns the instance count of the frame (rasterize_points.cu:119); on the sync-free forward the binning
    workspace is laid out for a capacity >= that count, which only the backward needs: an int that remembers it."""
    from diff_gaussian_rasterization import _backend as B
    n = B.NumRendered(8_837_593, 11_075_584)
    assert n == 8_837_593 and isinstance(n, int) and n + 1 == 8_837_594 and "%d" % n == "8837593"
    assert n.layout == 11_075_584 and B._layout_count(n) == 11_075_584
    m = B.NumRendered(123)
    assert m.layout == 123 and B._layout_count(m) == 123 and B._layout_count(77) == 77       # a plain int (two-stage path, C callers) is its own layout
    assert B._round_capacity(8_837_593) >= int(8_837_593 * 1.25) and B._round_capacity(0) == 1 << 16


def test_pose_matrix_remembers_its_inverse_and_is_a_plain_tensor_otherwise():
    """train_epilogue.PoseMatrix (what the launcher makes of Camera.world_view_transform): train.py:177-179's `.T.inverse()` is
    computed once and again after an in-place edit; `c2w[:3, :3] @ X` gives torch's product (on the host: torch's own matmul -- the
    streaming kernel is for device tensors, tests/test_train_epilogue_gpu.py); every other operation returns plain tensors."""
    import train_epilogue as T
    g = torch.Generator().manual_seed(3)
    w = torch.eye(4)
    w[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    w[3, :3] = torch.tensor([0.3, -1.0, 2.0])
    p = T.PoseMatrix.wrap(w.clone())
    assert isinstance(p, torch.Tensor) and torch.equal(p, w) and T.PoseMatrix.wrap(p) is p
    c2w = p.T.inverse()                                                     # train.py:177
    assert torch.equal(c2w, w.T.inverse()) and p.T.inverse() is c2w       # the same object: nothing is recomputed
    x = torch.randn(3, 5000, generator=g)
    assert torch.equal(c2w[:3, :3] @ x, w.T.inverse()[:3, :3] @ x)         # :178
    for r in (p + 1, p[:3, :3], p.T @ p, p.T.contiguous(), c2w * 2, torch.stack([p, p])):
        assert type(r) is torch.Tensor
    p.mul_(1.0)                                                             # an in-place edit invalidates the remembered inverse
    assert p.T.inverse() is not c2w and torch.equal(p.T.inverse(), c2w)


def test_learnt_pools_are_inherited_when_densification_changes_the_number_of_gaussians():
    """_backend._inherit_learnt: training changes P every 100 iterations (train.py:258-264); a new P at a resolution whose previous
    frames had between half and twice as many Gaussians takes over their capacity / mask / record needs, scaled up by the growth (never
    down), and the old shape's entries go -- no first-frame read-back per densification.  Outside that window nothing is inherited."""
    from diff_gaussian_rasterization import _backend as B
    keep = (dict(B._capacity), dict(B._mask_need), dict(B._staged_need), dict(B._recent_P), dict(B._stats))
    try:
        for d in (B._capacity, B._mask_need, B._staged_need, B._recent_P):
            d.clear()
        k0 = ("cuda:0", 100_000, 800, 800)
        B._inherit_learnt(k0)                                   # first shape at this resolution: nothing to inherit
        assert not B._capacity
        B._capacity[k0], B._mask_need[k0], B._staged_need[k0] = 3 << 16, 1000, 50_000
        k1 = ("cuda:0", 130_000, 800, 800)
        B._inherit_learnt(k1)
        assert k0 not in B._capacity and B._capacity[k1] >= int((3 << 16) * 1.3) and B._capacity[k1] % (1 << 16) == 0
        assert B._mask_need[k1] >= 1300 and B._staged_need[k1] >= 65_000 and k0 not in B._mask_need and k0 not in B._staged_need
        k2 = ("cuda:0", 90_000, 800, 800)                      # pruned: inherited as it is (a capacity above the count costs nothing)
        cap1 = B._capacity[k1]
        B._inherit_learnt(k2)
        assert B._capacity[k2] == cap1 and k1 not in B._capacity
        k3 = ("cuda:0", 30_000, 800, 800)                      # a third of it: another scene as far as the pools are concerned
        B._inherit_learnt(k3)
        assert k3 not in B._capacity
        k4 = ("cuda:0", 95_000, 400, 400)                      # another resolution: its own history
        B._inherit_learnt(k4)
        assert k4 not in B._capacity
    finally:
        for d, k in zip((B._capacity, B._mask_need, B._staged_need, B._recent_P, B._stats), keep):
            d.clear(); d.update(k)


def test_per_call_modes_are_per_thread_and_nest():
    """_backend.call_modes: the modes handed to the library in GofRasterArgs (ABI 12) -- 0 = the process default, 1 = on, -1 = off."""
    import threading
    from diff_gaussian_rasterization import _backend as B
    assert (B._mode_field("forward_exact"), B._mode_field("tight_tile_rects"), B._mode_field("integrate_pixel_pass")) == (0, 0, 0)
    seen = {}
    with B.call_modes(forward_exact=True):
        assert B._mode_field("forward_exact") == 1 and B._mode_field("tight_tile_rects") == 0
        t = threading.Thread(target=lambda: seen.update(other=B._mode_field("forward_exact")))
        t.start(); t.join()
        with B.call_modes(forward_exact=False, integrate_pixel_pass=True):
            assert (B._mode_field("forward_exact"), B._mode_field("integrate_pixel_pass")) == (-1, 1)
            with B.call_modes(tight_tile_rects=True):
                assert (B._mode_field("forward_exact"), B._mode_field("tight_tile_rects"), B._mode_field("integrate_pixel_pass")) == (-1, 1, 1)
        assert B._mode_field("forward_exact") == 1 and B._mode_field("integrate_pixel_pass") == 0
    assert seen["other"] == 0 and B._mode_field("forward_exact") == 0


def test_channel_slices_of_the_image_reach_the_backward_through_one_buffer():
    """RenderedImage: the four slices train.py takes of the (9, H, W) image (train.py:149-172) hand their gradients to the rasterizer's
    backward through ONE zero-filled buffer instead of autograd's zero-fill + add per slice.  With a stand-in for the rasterizer (no GPU
    here): the gradients are those of the plain tensor -- bit for bit when the slices are disjoint, to rounding when a whole-image use
    or overlapping slices make autograd / the slab add --, other operations return plain tensors, a retained graph can be walked twice."""
    class Stand(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.slab = dgr._GradSlab()
            ctx.set_materialize_grads(False)
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            return ctx.slab.deliver(g) * 2.0
    x = torch.randn(9, 6, 5, requires_grad=True)

    def image(plain):
        y = Stand.apply(x)
        return y if plain else dgr._as_rendered_image(y, getattr(y.grad_fn, "slab", None))      # (as diff_gaussian_rasterization.rasterize_gaussians does)

    def train_py_loss(img):
        return img[:3, :, :].abs().sum() + 3 * img[8, :, :].mean() + (img[3:6, :, :] ** 2).sum() + 0.5 * img[6, :, :].sum()
    want, = torch.autograd.grad(train_py_loss(image(True)), x)
    got, = torch.autograd.grad(train_py_loss(image(False)), x)
    assert torch.equal(got, want) and got[7].abs().max() == 0          # (channel 7: nobody used it -- zero, as with autograd's slices)

    def mixed(img):
        return img[:3].sum() + (img * 0.25).sum() + img[2].sum() + img[1:4].mean() + img[-1].sum()
    want, = torch.autograd.grad(mixed(image(True)), x)
    got, = torch.autograd.grad(mixed(image(False)), x)
    assert (got - want).abs().max() <= 1e-6
    img = image(False)
    assert type(img) is dgr.RenderedImage and type(img[:3]) is torch.Tensor and type(img + 1) is torch.Tensor and type(img.permute(1, 2, 0)) is torch.Tensor
    assert type(img[:, 2:4]) is torch.Tensor and type(img[::2]) is torch.Tensor        # not whole channels: autograd's own slices
    loss = train_py_loss(img)
    g1, = torch.autograd.grad(loss, x, retain_graph=True)
    g2, = torch.autograd.grad(loss, x)
    assert torch.equal(g1, g2)
    with torch.no_grad():
        assert type(image(False)) is torch.Tensor                     # nothing to differentiate: the plain tensor
