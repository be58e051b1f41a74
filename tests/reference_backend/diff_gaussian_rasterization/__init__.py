"""TEST INFRASTRUCTURE -- never the product: a ``diff_gaussian_rasterization`` module whose kernels are the REFERENCE's own
(submodules/diff-gaussian-rasterization/cuda_rasterizer/*.cu compiled for gfx950 where they lie by oracle/build_ref.sh ->
oracle/_ref/libgof_cudaref.so, default FMA contraction), behind the reference's Python surface (GaussianRasterizationSettings,
GaussianRasterizer.forward; diff_gaussian_rasterization/__init__.py:58-183 of the reference).  tests/test_trajectory_gpu.py runs the
reference's UNCHANGED train.py once on this module and once on the product and compares the training trajectories
(tests/reference_backend/run_with_reference_rasterizer.py puts this module in place of the product's).  Forward + backward only:
what train.py's render() needs."""
import ctypes as C
import os
import sys
from typing import NamedTuple

import torch
import torch.nn as nn

_TESTS = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _TESTS not in sys.path:
    sys.path.insert(0, _TESTS)
from oracle_binding import GofRasterArgs, ROOT          # noqa: E402

_SO = os.path.join(ROOT, "oracle", "_ref", "libgof_cudaref%s.so" % os.environ.get("GOF_REFERENCE_VARIANT", ""))
_L = C.CDLL(_SO)
_L.cudaref_create.restype = C.c_void_p
_L.cudaref_destroy.argtypes = [C.c_void_p]
_L.cudaref_forward.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs), C.c_void_p, C.c_void_p]
_L.cudaref_backward.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs)] + [C.c_void_p] * 12
IS_REFERENCE_BACKEND = True

# The launcher's simple_knn._C (distCUDA2 on HIP: scene initialisation, identical on both sides of the comparison) finds the shared
# library through `diff_gaussian_rasterization._backend`: the product's ctypes loader, imported here by file path under that name.
# The RASTERIZER of this module stays the reference's kernels (_Rasterize below never touches it).
import importlib.util as _ilu      # noqa: E402
_spec = _ilu.spec_from_file_location("diff_gaussian_rasterization._backend",
                                     os.path.join(ROOT, "gaussian-opacity-fields_amd", "diff_gaussian_rasterization", "_backend.py"))
_backend = _ilu.module_from_spec(_spec)
sys.modules["diff_gaussian_rasterization._backend"] = _backend
_spec.loader.exec_module(_backend)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    subpixel_offset: torch.Tensor
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _p(t):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


class _State:
    """one reference-side state (geometry / binning / image buffers) per forward call, freed after its backward or with the graph"""

    def __init__(self):
        self.h = C.c_void_p(_L.cudaref_create())

    def __del__(self):
        try:
            torch.cuda.synchronize()
            _L.cudaref_destroy(self.h)
        except Exception:
            pass


def _args(rs, keep):
    a = GofRasterArgs()
    P = int(keep["means3D"].shape[0])
    sh = keep["sh"]
    a.P, a.D, a.M, a.W, a.H = P, int(rs.sh_degree), (int(sh.shape[1]) if sh is not None and sh.numel() else 0), int(rs.image_width), int(rs.image_height)
    a.tan_fovx, a.tan_fovy, a.kernel_size, a.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.kernel_size), float(rs.scale_modifier)
    a.prefiltered, a.debug = 0, 0
    a.background = _p(keep["bg"]); a.means3D = _p(keep["means3D"]); a.shs = _p(sh); a.colors_precomp = _p(keep["colors"])
    a.opacities = _p(keep["opacities"]); a.scales = _p(keep["scales"]); a.rotations = _p(keep["rotations"])
    a.cov3D_precomp = _p(keep["cov3D"]); a.view2gaussian_precomp = _p(keep["v2g"])
    a.viewmatrix = _p(keep["view"]); a.projmatrix = _p(keep["proj"]); a.campos = _p(keep["campos"]); a.subpixel_offset = _p(keep["subpix"])
    a.shs_rest = None
    return a


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, rs):
        f = lambda t: None if t is None or t.numel() == 0 else t.detach().contiguous().float()      # noqa: E731
        keep = dict(means3D=f(means3D), sh=f(sh), colors=f(colors_precomp), opacities=f(opacities), scales=f(scales), rotations=f(rotations),
                    cov3D=f(cov3Ds_precomp), v2g=f(view2gaussian_precomp), bg=f(rs.bg), view=f(rs.viewmatrix), proj=f(rs.projmatrix),
                    campos=f(rs.campos), subpix=f(rs.subpixel_offset))
        dev = means3D.device
        P = int(means3D.shape[0])
        out = torch.zeros((9, int(rs.image_height), int(rs.image_width)), device=dev)
        radii = torch.zeros(P, dtype=torch.int32, device=dev)
        st = _State()
        a = _args(rs, keep)
        torch.cuda.synchronize()
        _L.cudaref_forward(st.h, C.byref(a), _p(out), _p(radii))
        ctx.st, ctx.keep, ctx.rs, ctx.radii = st, keep, rs, radii
        ctx.mark_non_differentiable(radii)
        return out, radii

    @staticmethod
    def backward(ctx, g_out, _g_radii):
        keep, rs = ctx.keep, ctx.rs
        dev = keep["means3D"].device
        P = int(keep["means3D"].shape[0])
        M = int(keep["sh"].shape[1]) if keep["sh"] is not None else 0
        z = lambda *s: torch.zeros(s, device=dev)      # noqa: E731
        g = dict(means2D=z(P, 3), colors=z(P, 3), opacity=z(P, 1), means3D=z(P, 3), cov3D=z(P, 6), sh=z(P, max(M, 1), 3), scales=z(P, 3),
                 rotations=z(P, 4), v2g=z(P, 10), conic=z(P, 4))
        a = _args(rs, keep)
        d = g_out.contiguous().float()
        torch.cuda.synchronize()
        _L.cudaref_backward(ctx.st.h, C.byref(a), _p(ctx.radii), _p(d), _p(g["means2D"]), _p(g["colors"]), _p(g["opacity"]), _p(g["means3D"]),
                            _p(g["cov3D"]), _p(g["sh"]) if M else None, _p(g["scales"]), _p(g["rotations"]), _p(g["v2g"]), _p(g["conic"]))
        ctx.st = None
        # (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, view2gaussian_precomp, rs)
        return (g["means3D"], g["means2D"], g["sh"] if keep["sh"] is not None else None, g["colors"] if keep["colors"] is not None else None,
                g["opacity"], g["scales"] if keep["scales"] is not None else None, g["rotations"] if keep["rotations"] is not None else None,
                g["cov3D"] if keep["cov3D"] is not None else None, g["v2g"] if keep["v2g"] is not None else None, None)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, view2gaussian_precomp=None):
        if hasattr(shs, "cat"):            # the launcher's SplitSH view of (_features_dc, _features_rest): this backend takes the concatenation
            shs = shs.cat()
        return _Rasterize.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, view2gaussian_precomp, self.raster_settings)
