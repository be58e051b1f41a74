"""ctypes binding of oracle/_ref/libgof_cudaref*.so (TEST INFRASTRUCTURE): the REFERENCE CUDA
rasterizer compiled for gfx950 by oracle/build_ref.sh.  Device memory via torch."""
import contextlib
import ctypes as C
import os
import sys

import numpy as np
import torch

from oracle_binding import GofRasterArgs, ROOT

REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def available(variant=""):
    return os.path.exists(os.path.join(REF_DIR, "libgof_cudaref%s.so" % variant))


_ELEM = {"depths": np.float32, "means2D": np.float32, "cov3D": np.float32, "view2gaussian": np.float32, "conic_opacity": np.float32,
         "rgb": np.float32, "clamped": np.uint8, "tiles_touched": np.uint32, "point_offsets": np.uint32, "point_list": np.uint32,
         "point_list_keys": np.uint64, "ranges": np.uint32, "final_T": np.float32, "n_contrib": np.uint32}


@contextlib.contextmanager
def _stdout_to_devnull():
    """The reference's integrateCUDA printf()s "ERROR: Maximal contributors are met..." once per saturated pixel
    (forward.cu:988): tens of thousands of lines that push pytest's summary out of the recorded tail.  Device printf goes to
    the C library's stdout (buffered when fd 1 is a pipe) as the runtime services the kernel's host calls, so fd 1 points at
    /dev/null from the launch until the stream is synchronised AND the C stream is flushed."""
    libc = C.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    null = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(null, 1)
        yield
        torch.cuda.synchronize()
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
        os.close(null)


class Reference:
    def __init__(self, scene_dev, variant=""):
        L = self.L = C.CDLL(os.path.join(REF_DIR, "libgof_cudaref%s.so" % variant))
        L.cudaref_create.restype = C.c_void_p
        L.cudaref_destroy.argtypes = [C.c_void_p]
        L.cudaref_forward.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs), C.c_void_p, C.c_void_p]
        L.cudaref_backward.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs)] + [C.c_void_p] * 12
        L.cudaref_integrate.argtypes = [C.c_void_p, C.POINTER(GofRasterArgs), C.c_int] + [C.c_void_p] * 5
        L.cudaref_fetch.restype = C.c_void_p
        L.cudaref_fetch.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
        self.h = C.c_void_p(L.cudaref_create())
        sd = self.sd = scene_dev
        self.P = sd["means3D"].shape[0]
        self.M = sd["shs"].shape[1]
        self.H, self.W = sd["H"], sd["W"]
        a = self.args = GofRasterArgs()
        a.P, a.D, a.M, a.W, a.H = self.P, sd["sh_degree"], self.M, self.W, self.H
        a.tan_fovx, a.tan_fovy, a.kernel_size, a.scale_modifier = sd["tanfovx"], sd["tanfovy"], sd["kernel_size"], sd["scale_modifier"]
        a.prefiltered, a.debug = 0, 0
        self.keep = {k: sd[k].contiguous() for k in ("bg", "means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "campos", "subpixel_offset")}
        k = self.keep
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        a.background = p(k["bg"]); a.means3D = p(k["means3D"]); a.shs = p(k["shs"]); a.colors_precomp = None
        a.opacities = p(k["opacities"]); a.scales = p(k["scales"]); a.rotations = p(k["rotations"])
        a.cov3D_precomp = None; a.view2gaussian_precomp = None
        a.viewmatrix = p(k["viewmatrix"]); a.projmatrix = p(k["projmatrix"]); a.campos = p(k["campos"]); a.subpixel_offset = p(k["subpixel_offset"])

    def __del__(self):
        try:
            self.L.cudaref_destroy(self.h)
        except Exception:
            pass

    def forward(self):
        dev = self.sd["means3D"].device
        self.out = torch.zeros((9, self.H, self.W), device=dev)
        self.radii = torch.zeros(self.P, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        self.R = self.L.cudaref_forward(self.h, C.byref(self.args), C.c_void_p(self.out.data_ptr()), C.c_void_p(self.radii.data_ptr()))
        return self.out.cpu().numpy(), self.radii.cpu().numpy()

    def time_forward_backward(self, dL_dev, iters=3):
        """fwd + bwd of the reference's own kernels with everything resident on the device (no host copies inside the timed part;
        the C wrapper synchronises after each of the two calls, as torch's binding would not -- the reference is timed generously):
        -> (fwd_ms, bwd_ms) medians.  bench.py's `reference_same_gpu` leg."""
        import time
        dev = self.sd["means3D"].device
        P, M = self.P, self.M
        z = lambda *s: torch.zeros(s, device=dev)   # noqa: E731
        out = z(9, self.H, self.W); radii = torch.zeros(P, dtype=torch.int32, device=dev)
        g = [z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 6), z(P, M, 3), z(P, 3), z(P, 4), z(P, 10), z(P, 4)]
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        tf, tb = [], []
        for _ in range(iters + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.R = self.L.cudaref_forward(self.h, C.byref(self.args), p(out), p(radii))
            t1 = time.perf_counter()
            self.L.cudaref_backward(self.h, C.byref(self.args), p(radii), p(dL_dev), *[p(t) for t in g])
            t2 = time.perf_counter()
            tf.append(1e3 * (t1 - t0)); tb.append(1e3 * (t2 - t1))
        return float(np.median(tf[1:])), float(np.median(tb[1:]))

    def fetch(self, name):
        n = C.c_longlong(0)
        ptr = self.L.cudaref_fetch(self.h, name.encode(), C.byref(n))
        if n.value < 0:
            raise KeyError(name)
        dt = np.dtype(_ELEM[name])
        host = np.empty(n.value, dtype=dt)
        if n.value:
            t = torch.empty(n.value * dt.itemsize, dtype=torch.uint8, device=self.sd["means3D"].device)
            # device-to-device copy through hipMemcpy via torch: build a tensor view by ctypes memmove is not possible; use hip
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), n.value * dt.itemsize, 3) == 0
            host = t.cpu().numpy().view(dt).copy()
        return host

    def backward(self, dL):
        dev = self.sd["means3D"].device
        P, M = self.P, self.M
        z = lambda *s: torch.zeros(s, device=dev)   # noqa: E731
        g = dict(means2D=z(P, 3), colors=z(P, 3), opacity=z(P, 1), means3D=z(P, 3), cov3D=z(P, 6), sh=z(P, M, 3), scales=z(P, 3),
                 rotations=z(P, 4), view2gaussian=z(P, 10))
        conic = z(P, 4)
        d = torch.from_numpy(np.ascontiguousarray(dL, dtype=np.float32)).to(dev)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        torch.cuda.synchronize()
        self.L.cudaref_backward(self.h, C.byref(self.args), p(self.radii), p(d), p(g["means2D"]), p(g["colors"]), p(g["opacity"]),
                                p(g["means3D"]), p(g["cov3D"]), p(g["sh"]), p(g["scales"]), p(g["rotations"]), p(g["view2gaussian"]), p(conic))
        return {k: v.cpu().numpy() for k, v in g.items()}

    def integrate(self, points3D):
        dev = self.sd["means3D"].device
        pts = torch.from_numpy(np.ascontiguousarray(points3D, dtype=np.float32)).to(dev)
        PN = pts.shape[0]
        out = torch.zeros((9, self.H, self.W), device=dev)
        alpha = torch.ones(PN, device=dev)
        col = torch.zeros((PN, 3), device=dev)
        radii = torch.zeros(self.P, dtype=torch.int32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        torch.cuda.synchronize()
        with _stdout_to_devnull():
            self.R = self.L.cudaref_integrate(self.h, C.byref(self.args), PN, p(pts), p(out), p(alpha), p(col), p(radii))
        return out.cpu().numpy(), alpha.cpu().numpy(), col.cpu().numpy(), radii.cpu().numpy()
