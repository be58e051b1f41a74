"""numpy + ctypes binding of oracle/_ref/libgof_cudaref_host.so (TEST INFRASTRUCTURE): the REFERENCE's own CUDA rasterizer compiled
as host code on top of tests/hipemu by tests/hipemu/build_ref_host.py -- the yardstick the CPU suite pins the oracle with when no GPU
is at hand (same C wrapper, oracle/ref_capi.cpp, as the gfx950 builds that tests/reference_binding.py drives on a GPU box)."""
import contextlib
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import build_ref_host  # noqa: E402
from oracle_binding import GofRasterArgs  # noqa: E402

_ELEM = {"depths": np.float32, "means2D": np.float32, "cov3D": np.float32, "view2gaussian": np.float32, "conic_opacity": np.float32,
         "rgb": np.float32, "clamped": np.uint8, "tiles_touched": np.uint32, "point_offsets": np.uint32, "point_list": np.uint32,
         "point_list_keys": np.uint64, "ranges": np.uint32, "final_T": np.float32, "n_contrib": np.uint32}


def available():
    return build_ref_host.build() is not None


@contextlib.contextmanager
def _stdout_to_devnull():
    """the reference printf()s per saturated pixel in integrateCUDA (forward.cu:988)"""
    libc = C.CDLL(None)
    sys.stdout.flush(); libc.fflush(None)
    saved = os.dup(1); null = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(null, 1)
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1); os.close(saved); os.close(null)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return C.c_void_p(a.ctypes.data)


class ReferenceOnHost:
    def __init__(self, sc):
        L = self.L = C.CDLL(build_ref_host.build())
        L.cudaref_create.restype = C.c_void_p
        L.cudaref_destroy.argtypes = [C.c_void_p]
        L.cudaref_forward.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs), C.c_void_p, C.c_void_p]
        L.cudaref_backward.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs)] + [C.c_void_p] * 12
        L.cudaref_integrate.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs), C.c_int] + [C.c_void_p] * 5
        L.cudaref_fetch.restype = C.c_void_p
        L.cudaref_fetch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
        self.h = C.c_void_p(L.cudaref_create())
        k = self.keep = {n: _f32(sc[n]) for n in ("bg", "means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "campos", "subpixel_offset")}
        self.P = k["means3D"].shape[0]; self.M = k["shs"].shape[1]; self.H, self.W = int(sc["H"]), int(sc["W"])
        a = self.args = GofRasterArgs()
        a.P, a.D, a.M, a.W, a.H = self.P, int(sc["sh_degree"]), self.M, self.W, self.H
        a.tan_fovx, a.tan_fovy, a.kernel_size, a.scale_modifier = float(sc["tanfovx"]), float(sc["tanfovy"]), float(sc["kernel_size"]), float(sc["scale_modifier"])
        a.prefiltered, a.debug = 0, 0
        a.background = _p(k["bg"]); a.means3D = _p(k["means3D"]); a.shs = _p(k["shs"]); a.colors_precomp = None
        a.opacities = _p(k["opacities"]); a.scales = _p(k["scales"]); a.rotations = _p(k["rotations"])
        a.cov3D_precomp = None; a.view2gaussian_precomp = None
        a.viewmatrix = _p(k["viewmatrix"]); a.projmatrix = _p(k["projmatrix"]); a.campos = _p(k["campos"]); a.subpixel_offset = _p(k["subpixel_offset"])

    def __del__(self):
        try:
            self.L.cudaref_destroy(self.h)
        except Exception:
            pass

    def forward(self):
        self.out = np.zeros((9, self.H, self.W), np.float32)
        self.radii = np.zeros(self.P, np.int32)
        self.R = self.L.cudaref_forward(self.h, C.byref(self.args), _p(self.out), _p(self.radii))
        return self.out, self.radii

    def fetch(self, name):
        n = C.c_longlong(0)
        ptr = self.L.cudaref_fetch(self.h, name.encode(), C.byref(n))
        if n.value < 0:
            raise KeyError(name)
        dt = np.dtype(_ELEM[name])
        if not n.value:
            return np.empty(0, dt)
        buf = (C.c_char * (n.value * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    def backward(self, dL):
        P, M = self.P, self.M
        z = lambda *s: np.zeros(s, np.float32)   # noqa: E731
        g = dict(means2D=z(P, 3), colors=z(P, 3), opacity=z(P, 1), means3D=z(P, 3), cov3D=z(P, 6), sh=z(P, M, 3), scales=z(P, 3),
                 rotations=z(P, 4), view2gaussian=z(P, 10))
        conic = z(P, 4)
        d = _f32(dL)
        self.L.cudaref_backward(self.h, C.byref(self.args), _p(self.radii), _p(d), _p(g["means2D"]), _p(g["colors"]), _p(g["opacity"]),
                                _p(g["means3D"]), _p(g["cov3D"]), _p(g["sh"]), _p(g["scales"]), _p(g["rotations"]), _p(g["view2gaussian"]), _p(conic))
        return g

    def integrate(self, points3D):
        pts = _f32(points3D); PN = pts.shape[0]
        out = np.zeros((9, self.H, self.W), np.float32); alpha = np.ones(PN, np.float32); col = np.zeros((PN, 3), np.float32)
        radii = np.zeros(self.P, np.int32)
        with _stdout_to_devnull():
            self.R = self.L.cudaref_integrate(self.h, C.byref(self.args), PN, _p(pts), _p(out), _p(alpha), _p(col), _p(radii))
        return out, alpha, col, radii
