"""ctypes binding of oracle/liboracle_gof.so (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle_gof.so")


class GofRasterArgs(C.Structure):
    """Mirror of GofRasterArgs in include/gof_hip.h."""
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("kernel_size", C.c_float), ("scale_modifier", C.c_float),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("view2gaussian_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("campos", C.c_void_p), ("subpixel_offset", C.c_void_p), ("shs_rest", C.c_void_p),
        ("forward_exact", C.c_int32), ("tight_tile_rects", C.c_int32), ("integrate_pixel_pass", C.c_int32), ("reserved0", C.c_int32),   # shs_rest and the per-call modes: product only, NULL / 0 here
    ]


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".inc", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "gof_hip.h"))
    if (not force and os.path.exists(ORACLE_SO)
            and os.path.getmtime(ORACLE_SO) >= max(os.path.getmtime(s) for s in srcs)):
        return ORACLE_SO
    subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle_gof.so"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.gofref_last_error.restype = C.c_char_p
        L.gofref_expf.restype = C.c_float
        L.gofref_expf.argtypes = [C.c_float]
        L.gofref_forward.argtypes = [C.POINTER(GofRasterArgs), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.gofref_backward.argtypes = [C.POINTER(GofRasterArgs), C.c_void_p] + [C.c_void_p] * 10
        L.gofref_preprocess_backward.argtypes = [C.POINTER(GofRasterArgs), C.c_void_p] + [C.c_void_p] * 6
        L.gofref_integrate.argtypes = [C.POINTER(GofRasterArgs), C.c_int32] + [C.c_void_p] * 5 + [C.POINTER(C.c_void_p)]
        L.gofref_mark_visible.argtypes = [C.c_int32] + [C.c_void_p] * 4
        L.gofref_free.argtypes = [C.c_void_p]
        L.gofref_num_rendered.restype = C.c_uint32
        L.gofref_num_rendered.argtypes = [C.c_void_p]
        L.gofref_num_integrated.restype = C.c_uint32
        L.gofref_num_integrated.argtypes = [C.c_void_p]
        L.gofref_fetch.restype = C.c_int64
        L.gofref_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
        L.gofref_mtets.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 4 + [C.POINTER(C.c_int64)] * 2 + [C.c_void_p] * 5 + [C.c_int64] * 2
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


_DTYPES = {
    "depths": np.float32, "means2D": np.float32, "cov3D": np.float32, "view2gaussian": np.float32,
    "conic_opacity": np.float32, "rgb": np.float32, "clamped": np.uint8, "radii": np.int32,
    "tiles_touched": np.uint32, "point_offsets": np.uint32, "point_list_keys_unsorted": np.uint64,
    "point_list_keys": np.uint64, "point_list_unsorted": np.uint32, "point_list": np.uint32,
    "ranges": np.uint32, "final_T": np.float32, "n_contrib": np.uint32, "tile_walked": np.uint32,
    "p_depths": np.float32, "points2D": np.float32, "p_tiles_touched": np.uint32, "p_list": np.uint32,
    "p_ranges": np.uint32, "p_keys": np.uint64,
}


class OracleScene:
    """Holds contiguous float32 host copies of a scene dict and the matching GofRasterArgs."""

    def __init__(self, scene, colors_precomp=None, cov3D_precomp=None, view2gaussian_precomp=None,
                 use_shs=True, prefiltered=False):
        s = scene
        self.keep = {}
        k = self.keep
        k["bg"] = _f32(s["bg"]); k["means3D"] = _f32(s["means3D"])
        k["opacities"] = _f32(s["opacities"]).reshape(-1)
        k["scales"] = _f32(s.get("scales")); k["rotations"] = _f32(s.get("rotations"))
        k["shs"] = _f32(s["shs"]) if (use_shs and colors_precomp is None) else None
        k["colors_precomp"] = _f32(colors_precomp)
        k["cov3D_precomp"] = _f32(cov3D_precomp)
        k["view2gaussian_precomp"] = _f32(view2gaussian_precomp)
        k["viewmatrix"] = _f32(s["viewmatrix"]); k["projmatrix"] = _f32(s["projmatrix"])
        k["campos"] = _f32(s["campos"]); k["subpixel_offset"] = _f32(s["subpixel_offset"])
        P = k["means3D"].shape[0]
        M = k["shs"].shape[1] if k["shs"] is not None else 0
        a = GofRasterArgs()
        a.P, a.D, a.M, a.W, a.H = P, int(s["sh_degree"]), M, int(s["W"]), int(s["H"])
        a.tan_fovx, a.tan_fovy = float(s["tanfovx"]), float(s["tanfovy"])
        a.kernel_size, a.scale_modifier = float(s["kernel_size"]), float(s["scale_modifier"])
        a.prefiltered, a.debug = int(prefiltered), 0
        a.background = _p(k["bg"]); a.means3D = _p(k["means3D"]); a.shs = _p(k["shs"])
        a.colors_precomp = _p(k["colors_precomp"]); a.opacities = _p(k["opacities"])
        a.scales = _p(k["scales"]); a.rotations = _p(k["rotations"])
        a.cov3D_precomp = _p(k["cov3D_precomp"]); a.view2gaussian_precomp = _p(k["view2gaussian_precomp"])
        a.viewmatrix = _p(k["viewmatrix"]); a.projmatrix = _p(k["projmatrix"])
        a.campos = _p(k["campos"]); a.subpixel_offset = _p(k["subpixel_offset"])
        self.args = a
        self.P, self.M, self.W, self.H = P, M, a.W, a.H
        self.state = None

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + lib().gofref_last_error().decode())

    def free(self):
        if self.state:
            lib().gofref_free(self.state)
            self.state = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def forward(self):
        self.free()
        out = np.zeros((9, self.H, self.W), dtype=np.float32)
        radii = np.zeros(self.P, dtype=np.int32)
        st = C.c_void_p()
        self._check(lib().gofref_forward(C.byref(self.args), _p(out), _p(radii), C.byref(st)))
        self.state = st
        return out, radii

    def num_rendered(self):
        return int(lib().gofref_num_rendered(self.state))

    def num_integrated(self):
        return int(lib().gofref_num_integrated(self.state))

    def fetch(self, name):
        n = lib().gofref_fetch(self.state, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        arr = np.zeros(int(n), dtype=_DTYPES[name])
        if n:
            lib().gofref_fetch(self.state, name.encode(), _p(arr), arr.nbytes)
        return arr

    def backward(self, dL_dout):
        P, M = self.P, self.M
        g = dict(
            means2D=np.zeros((P, 3), np.float32), colors=np.zeros((P, 3), np.float32), opacity=np.zeros((P, 1), np.float32),
            means3D=np.zeros((P, 3), np.float32), cov3D=np.zeros((P, 6), np.float32), sh=np.zeros((P, M, 3), np.float32),
            scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32), view2gaussian=np.zeros((P, 10), np.float32))
        d = _f32(dL_dout)
        self._check(lib().gofref_backward(C.byref(self.args), self.state, _p(d), _p(g["means2D"]), _p(g["colors"]), _p(g["opacity"]),
                                          _p(g["means3D"]), _p(g["cov3D"]), _p(g["sh"]) if M else None, _p(g["scales"]),
                                          _p(g["rotations"]), _p(g["view2gaussian"])))
        return g

    def preprocess_backward(self, dL_dview2gaussian, dL_dcolors):
        """K9 alone on given inputs -> dict(means3D, sh, scales, rotations)."""
        P, M = self.P, self.M
        g = dict(means3D=np.zeros((P, 3), np.float32), sh=np.zeros((P, M, 3), np.float32),
                 scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
        dv = _f32(dL_dview2gaussian); dc = _f32(dL_dcolors)
        self._check(lib().gofref_preprocess_backward(C.byref(self.args), self.state, _p(dv), _p(dc), _p(g["means3D"]),
                                                     _p(g["sh"]) if M else None, _p(g["scales"]), _p(g["rotations"])))
        return g

    def integrate(self, points3D):
        self.free()
        pts = _f32(points3D)
        PN = pts.shape[0]
        out = np.zeros((9, self.H, self.W), dtype=np.float32)
        alpha = np.ones(PN, dtype=np.float32)
        color = np.zeros((PN, 3), dtype=np.float32)
        radii = np.zeros(self.P, dtype=np.int32)
        st = C.c_void_p()
        self._check(lib().gofref_integrate(C.byref(self.args), PN, _p(pts), _p(out), _p(alpha), _p(color), _p(radii), C.byref(st)))
        self.state = st
        return out, alpha, color, radii


def mark_visible(means3D, viewmatrix, projmatrix):
    m = _f32(means3D); v = _f32(viewmatrix); p = _f32(projmatrix)
    out = np.zeros(m.shape[0], dtype=np.uint8)
    lib().gofref_mark_visible(m.shape[0], _p(m), _p(v), _p(p), _p(out))
    return out.astype(bool)


def marching_tets(vertices, tets, sdf, scales):
    v = _f32(vertices); t = np.ascontiguousarray(tets, dtype=np.int64); s = _f32(sdf).reshape(-1); sc = _f32(scales).reshape(-1)
    ne, nf = C.c_int64(), C.c_int64()
    L = lib()
    rc = L.gofref_mtets(v.shape[0], t.shape[0], _p(t), _p(v), _p(s), _p(sc), C.byref(ne), C.byref(nf), None, None, None, None, None, 0, 0)
    assert rc == 0
    E, F = ne.value, nf.value
    ids = np.zeros((E, 2), np.int64); pos = np.zeros((E, 2, 3), np.float32); esdf = np.zeros((E, 2), np.float32)
    esc = np.zeros((E, 2), np.float32); faces = np.zeros((F, 3), np.int64)
    rc = L.gofref_mtets(v.shape[0], t.shape[0], _p(t), _p(v), _p(s), _p(sc), C.byref(ne), C.byref(nf), _p(ids), _p(pos), _p(esdf), _p(esc), _p(faces), E, F)
    assert rc == 0
    return ids, pos, esdf, esc, faces


def expf(x):
    return float(lib().gofref_expf(float(x)))
