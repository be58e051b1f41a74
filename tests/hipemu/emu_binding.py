"""numpy + ctypes binding of tests/hipemu/_build/libgof_hip_emu.so -- the product's kernel SOURCES compiled for the host (see
tests/hipemu/include/hip/hip_runtime.h) behind the same C ABI (include/gof_hip.h).  TEST INFRASTRUCTURE: lets the CPU test suite run
the kernels' logic against the oracle without a GPU.  The product (diff_gaussian_rasterization/_backend.py) refuses host tensors and
never sees this library; the signature table is the product's own (``_backend._load``), applied to the emulated library."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import build_emu  # noqa: E402

_libs = {}


def load(asan=False, extra_flags=(), tag=""):
    """Build (if stale) and load the emulated library with the product's ctypes signatures."""
    key = (asan, tuple(extra_flags), tag)
    if key not in _libs:
        path = build_emu.build(asan=asan, extra_flags=extra_flags, tag=tag)
        from diff_gaussian_rasterization import _backend as B
        keep = B.LIB_PATH
        try:
            B.LIB_PATH = path
            _libs[key] = B._load()
        finally:
            B.LIB_PATH = keep
    return _libs[key]


GUARD = 4096
_guards = []          # (what, guard view): every workspace handed to the library has GUARD bytes of 0xA5 behind its stated size


def _aligned(nbytes, dtype=np.uint8, align=256, what="workspace"):
    raw = np.zeros(int(nbytes) + align + GUARD, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    g = raw[off + int(nbytes):off + int(nbytes) + GUARD]
    g[:] = 0xA5
    _guards.append((what, int(nbytes), g))
    if len(_guards) > 256:
        del _guards[:128]
    return raw[off:off + int(nbytes)].view(dtype)


def guards_intact():
    """-> list of (what, size) of the workspaces allocated so far whose guard bytes were overwritten (a write past the stated size)"""
    return [(w, n) for w, n, g in _guards if not (g == 0xA5).all()]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None or a.size == 0 else C.c_void_p(a.ctypes.data)


_FETCH_DTYPES = None


class EmuScene:
    """One synthetic scene (synthetic_scenes.py dict) through the emulated library: forward(), backward(dL), fetch(name)."""

    def __init__(self, sc, lib=None, exact=False, tight=False, **over):
        """exact: the forward blend's verification mode (gof_set_forward_exact) for this object's forward calls;
        tight: the opt-in tile lists (gof_set_tight_tile_rects)"""
        from diff_gaussian_rasterization import _backend as B
        self.B = B
        self.lib = lib or load()
        self.exact = bool(exact)
        self.tight = bool(tight)
        self.sc = sc
        k = self.keep = {}
        k["bg"] = _f32(sc["bg"]); k["means3D"] = _f32(sc["means3D"]); k["opacity"] = _f32(sc["opacities"])
        k["scales"] = _f32(over.get("scales", sc["scales"])); k["rotations"] = _f32(over.get("rotations", sc["rotations"]))
        k["colors"] = _f32(over["colors_precomp"]) if "colors_precomp" in over else None
        k["sh"] = None if "colors_precomp" in over else _f32(sc["shs"])
        k["cov3D"] = _f32(over["cov3D_precomp"]) if "cov3D_precomp" in over else None
        k["v2g"] = _f32(over["view2gaussian_precomp"]) if "view2gaussian_precomp" in over else None
        k["view"] = _f32(sc["viewmatrix"]); k["proj"] = _f32(sc["projmatrix"]); k["campos"] = _f32(sc["campos"])
        k["subpix"] = _f32(sc["subpixel_offset"])
        self.P = int(k["means3D"].shape[0]); self.W = int(sc["W"]); self.H = int(sc["H"])
        self.M = int(k["sh"].shape[1]) if k["sh"] is not None else 0
        a = self.args = B.GofRasterArgs()
        a.P, a.D, a.M, a.W, a.H = self.P, int(sc["sh_degree"]), self.M, self.W, self.H
        a.tan_fovx, a.tan_fovy = float(sc["tanfovx"]), float(sc["tanfovy"])
        a.kernel_size, a.scale_modifier = float(sc["kernel_size"]), float(sc["scale_modifier"])
        a.prefiltered, a.debug = int(bool(over.get("prefiltered", False))), int(bool(over.get("debug", False)))
        a.background = _p(k["bg"]); a.means3D = _p(k["means3D"]); a.shs = _p(k["sh"]); a.colors_precomp = _p(k["colors"])
        a.opacities = _p(k["opacity"]); a.scales = _p(k["scales"]); a.rotations = _p(k["rotations"])
        a.cov3D_precomp = _p(k["cov3D"]); a.view2gaussian_precomp = _p(k["v2g"])
        a.viewmatrix = _p(k["view"]); a.projmatrix = _p(k["proj"]); a.campos = _p(k["campos"]); a.subpixel_offset = _p(k["subpix"])
        a.shs_rest = None

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("libgof_hip_emu: " + self.lib.gof_last_error().decode(errors="replace"))

    def mode_arrays(self):
        """what gpu_common.assert_fast_mode_matches_exact compares"""
        return dict(color=self.color, radii=self.radii, final_T=self.fetch("final_T"), n_contrib=self.fetch("n_contrib"),
                    contrib_hash=self.fetch("contrib_hash"), tile_cost=self.fetch("tile_cost"))

    def forward(self, mask_subchunks=None):
        """mask_subchunks: size the binning workspace's contributor-mask pool for that many sub-chunks (gof_binning_bytes_for)
        instead of the worst case -- allocated at exactly that size, guard bytes behind it"""
        lib = self.lib
        lib.gof_set_forward_exact(1 if self.exact else 0)
        lib.gof_set_tight_tile_rects(1 if self.tight else 0)
        self.geom = _aligned(lib.gof_geom_bytes_forward(self.P), what="geom"); self.img = _aligned(lib.gof_image_bytes(self.W, self.H), what="image")      # (geometry as the binding sizes it for a forward: without the query's 16 B per Gaussian; guard bytes behind it)
        self.radii = np.zeros(self.P, np.int32)
        n = C.c_uint32(0)
        lib.gof_set_tight_tile_rects(1 if self.tight else 0)
        self._check(lib.gof_forward_prepare(C.byref(self.args), _p(self.geom), self.geom.size, _p(self.img), self.img.size, _p(self.radii), C.byref(n), None))
        self.R = int(n.value)
        nb = lib.gof_binning_bytes(self.R, self.W, self.H) if mask_subchunks is None else lib.gof_binning_bytes_for(self.R, self.W, self.H, int(mask_subchunks))
        self.binning = _aligned(nb, what="binning")
        self.color = np.zeros((9, self.H, self.W), np.float32)
        self._check(lib.gof_forward_render(C.byref(self.args), self.R, _p(self.radii), _p(self.geom), self.geom.size, _p(self.binning), self.binning.size,
                                           _p(self.img), self.img.size, _p(self.color), None))
        return self.color, self.radii

    def backward(self, dL, full_scratch=False):
        lib, P, M = self.lib, self.P, self.M
        dl = _f32(dL)
        g = {"means2D": np.zeros((P, 3), np.float32), "colors": np.zeros((P, 3), np.float32), "opacity": np.zeros((P, 1), np.float32),
             "means3D": np.zeros((P, 3), np.float32), "cov3D": np.zeros((P, 6), np.float32), "sh": np.zeros((P, max(M, 1), 3), np.float32),
             "scales": np.zeros((P, 3), np.float32), "rotations": np.zeros((P, 4), np.float32), "view2gaussian": np.zeros((P, 10), np.float32)}
        for v in g.values():
            v.fill(np.nan)          # the library must write every element it owns
        g["cov3D"].fill(0)
        # the record pool sized for exactly what the forward staged (gof_backward_query), as the product's binding does; the guard
        # bytes behind it catch a pool overrun.  full_scratch=True: the worst case, a record per instance
        if full_scratch:
            nscratch = lib.gof_backward_scratch_bytes(P, self.R)
        else:
            q = (C.c_uint32 * 3)()
            self._check(lib.gof_backward_query(C.byref(self.args), self.R, self.binning.size, _p(self.img), self.img.size, q, None))
            self.staged, self.masks_requested, self.masks_held = int(q[0]), int(q[1]), int(q[2])
            if self.masks_requested > self.masks_held:
                raise RuntimeError("mask pool too small: %d sub-chunks requested, %d held" % (self.masks_requested, self.masks_held))
            nscratch = lib.gof_backward_scratch_bytes_for(P, self.R, self.staged)
        scratch = _aligned(nscratch, what="backward scratch")
        self._check(lib.gof_backward(C.byref(self.args), self.R, _p(self.radii), _p(self.geom), self.geom.size, _p(self.binning), self.binning.size,
                                     _p(self.img), self.img.size, _p(dl), _p(g["means2D"]), _p(g["colors"]), _p(g["opacity"]), _p(g["means3D"]), None,
                                     _p(g["sh"]) if M else None, None, _p(g["scales"]), _p(g["rotations"]), _p(g["view2gaussian"]), _p(scratch), nscratch, None))
        if not M:
            g["sh"] = np.zeros((P, 0, 3), np.float32)
        return g

    def fetch(self, name):
        import torch
        B = self.B
        dtype, per = B._FETCH[name]
        npdt = torch.empty(0, dtype=dtype).numpy().dtype
        P, HW = self.P, self.H * self.W
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        count = {"point_list": self.R, "point_list_keys": self.R, "ranges": 2 * T, "point_ranges": 2 * T, "final_T": 4 * HW, "n_contrib": 2 * HW,
                 "contrib_pairs": T, "contrib_hash": T, "tile_cost": T, "tile_order": 8 * ((T + 7) // 8 + 128), "tile_order_bw": 8 * ((T + 7) // 8 + 128),
                 "tile_queue": 64}.get(name, P * per)
        out = np.zeros(count, npdt)
        n = self.lib.gof_debug_fetch(name.encode(), C.byref(self.args), self.R, _p(self.geom), _p(self.binning), _p(self.img), _p(out), out.nbytes, None)
        if n < 0:
            self._check(int(n))
        return out

    def integrate(self, points3D):
        """-> (out_color [9,H,W], alpha_integrated [N], color_integrated [N,3], radii): gof_integrate_view + prepare_points + points."""
        lib = self.lib
        pts = _f32(points3D); PN = int(pts.shape[0])
        self.geom = _aligned(lib.gof_geom_bytes(self.P), what="geom"); self.img = _aligned(lib.gof_image_bytes(self.W, self.H), what="image")
        self.radii = np.zeros(self.P, np.int32)
        n = C.c_uint32(0)
        lib.gof_set_tight_tile_rects(1 if self.tight else 0)
        self._check(lib.gof_integrate_prepare(C.byref(self.args), _p(self.geom), self.geom.size, _p(self.img), self.img.size, _p(self.radii), C.byref(n), None))
        self.R = int(n.value)
        self.binning = _aligned(lib.gof_binning_bytes(self.R, self.W, self.H), what="binning")
        base = np.zeros((9, self.H, self.W), np.float32)
        self._check(lib.gof_integrate_view(C.byref(self.args), self.R, _p(self.radii), _p(self.geom), self.geom.size, _p(self.binning), self.binning.size,
                                           _p(self.img), self.img.size, _p(base), None))
        pws = _aligned(lib.gof_point_bytes(PN), what="point ws")
        ni = C.c_uint32(0)
        self._check(lib.gof_integrate_prepare_points(C.byref(self.args), PN, _p(pts), _p(pws), pws.size, C.byref(ni), None))
        self.NI = int(ni.value)
        pbin = _aligned(lib.gof_point_binning_bytes(self.NI, self.W, self.H), what="point binning")
        alpha = np.ones(PN, np.float32); colp = np.zeros((PN, 3), np.float32)
        self._check(lib.gof_integrate_points(C.byref(self.args), self.R, PN, self.NI, _p(self.geom), self.geom.size, _p(self.binning), self.binning.size,
                                             _p(self.img), self.img.size, _p(pws), pws.size, _p(pbin), pbin.size, _p(base), _p(base), _p(alpha), _p(colp), None))
        return base, alpha, colp, self.radii

    def forward_fused(self, capacity, guard=4096):
        """gof_forward_fused (the sync-free forward: binning workspace sized for `capacity` instances, the count read on the device).
        The binning workspace is allocated at EXACTLY gof_binning_bytes(capacity) with `guard` bytes of 0xA5 behind it.
        -> (rc, true instance count, guard intact?)"""
        lib = self.lib
        self.geom = _aligned(lib.gof_geom_bytes_forward(self.P), what="geom"); self.img = _aligned(lib.gof_image_bytes(self.W, self.H), what="image")
        nb = int(lib.gof_binning_bytes(int(capacity), self.W, self.H))
        raw = _aligned(nb + guard)
        raw[nb:] = 0xA5
        self.binning = raw[:nb]
        self.radii = np.zeros(self.P, np.int32)
        self.color = np.zeros((9, self.H, self.W), np.float32)
        pinned = np.zeros(4, np.uint32)
        self.usage_words = np.full(66, 0xDEADBEEF, np.uint32)      # GOF_USAGE_WORDS: the frame's raw pool counters, stored by the forward's last kernel
        lib.gof_set_forward_exact(1 if self.exact else 0)
        lib.gof_set_tight_tile_rects(1 if self.tight else 0)
        rc = lib.gof_forward_fused(C.byref(self.args), int(capacity), _p(self.geom), self.geom.size, _p(self.binning), nb, _p(self.img), self.img.size,
                                   _p(self.radii), _p(self.color), _p(pinned), _p(self.usage_words), None)
        self.R = int(capacity)           # the layout size: what fetch() / backward() have to be given on this path
        return rc, int(pinned[0]), bool((raw[nb:] == 0xA5).all())

    def usage_decoded(self):
        """gof_usage_decode of the words the fused forward left: (staged entries, mask sub-chunks requested, held)"""
        q = (C.c_uint32 * 3)()
        self._check(self.lib.gof_usage_decode(_p(self.usage_words), self.R, self.W, self.H, C.c_size_t(self.binning.size), q))
        return int(q[0]), int(q[1]), int(q[2])

    def integrate_view(self):
        """the Gaussian half of the opacity-field query (binning + pixel pass), kept on the object: -> base image [9,H,W]"""
        lib = self.lib
        self.geom = _aligned(lib.gof_geom_bytes(self.P), what="geom"); self.img = _aligned(lib.gof_image_bytes(self.W, self.H), what="image")
        self.radii = np.zeros(self.P, np.int32)
        n = C.c_uint32(0)
        lib.gof_set_tight_tile_rects(1 if self.tight else 0)
        self._check(lib.gof_integrate_prepare(C.byref(self.args), _p(self.geom), self.geom.size, _p(self.img), self.img.size, _p(self.radii), C.byref(n), None))
        self.R = int(n.value)
        self.binning = _aligned(lib.gof_binning_bytes(self.R, self.W, self.H), what="binning")
        self.base = np.zeros((9, self.H, self.W), np.float32)
        self._check(lib.gof_integrate_view(C.byref(self.args), self.R, _p(self.radii), _p(self.geom), self.geom.size, _p(self.binning), self.binning.size,
                                           _p(self.img), self.img.size, _p(self.base), None))
        return self.base

    def pack_geom(self):
        """keep only what the point pass reads of the geometry workspace (the per-view cache of a mesh extraction)"""
        lib = self.lib
        self.packed = _aligned(lib.gof_integrate_packed_geom_bytes(self.P), what="packed geom")
        self._check(lib.gof_integrate_pack_geom(C.byref(self.args), _p(self.geom), self.geom.size, _p(self.packed), self.packed.size, None))
        return self.packed

    def integrate_points(self, points3D, mode="plain", alpha_min=None, color_min=None):
        """the point half on the view prepared by integrate_view(): mode 'plain' | 'packed' (reads pack_geom()'s buffer) |
        'min' / 'min_packed' (min over views fused into the store: alpha_min / color_min updated in place)"""
        lib = self.lib
        pts = _f32(points3D); PN = int(pts.shape[0])
        pws = _aligned(lib.gof_point_bytes(PN), what="point ws")
        ni = C.c_uint32(0)
        self._check(lib.gof_integrate_prepare_points(C.byref(self.args), PN, _p(pts), _p(pws), pws.size, C.byref(ni), None))
        NI = int(ni.value)
        pbin = _aligned(lib.gof_point_binning_bytes(NI, self.W, self.H), what="point binning")
        use_packed = mode.endswith("packed")
        g = self.packed if use_packed else self.geom
        if mode.startswith("min"):
            self._check(lib.gof_integrate_points_min(C.byref(self.args), self.R, PN, NI, 1 if use_packed else 0, _p(g), g.size, _p(self.binning), self.binning.size,
                                                     _p(self.img), self.img.size, _p(pws), pws.size, _p(pbin), pbin.size, _p(self.base), None,
                                                     _p(alpha_min), None if color_min is None else _p(color_min), None))
            return alpha_min, color_min
        out = np.zeros((9, self.H, self.W), np.float32); alpha = np.ones(PN, np.float32); colp = np.zeros((PN, 3), np.float32)
        fn = lib.gof_integrate_points_packed if use_packed else lib.gof_integrate_points
        self._check(fn(C.byref(self.args), self.R, PN, NI, _p(g), g.size, _p(self.binning), self.binning.size, _p(self.img), self.img.size,
                       _p(pws), pws.size, _p(pbin), pbin.size, _p(self.base), _p(out), _p(alpha), _p(colp), None))
        return out, alpha, colp
