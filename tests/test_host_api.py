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
