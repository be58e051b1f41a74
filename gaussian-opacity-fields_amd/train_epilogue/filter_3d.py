"""Drop-in for ``GaussianModel.compute_3D_filter`` (reference scene/gaussian_model.py:262-311) on ``gof_compute_3d_filter``:
one launch over (points x all cameras) instead of ~25 torch kernels and two host-synchronising mask-index ops per camera.
launch/run_reference_script.py rebinds the method on the reference's GaussianModel class."""
import ctypes as C

import numpy as np
import torch

from . import _backend as B

lib = B.lib
lib.gof_filter3d_ws_bytes.restype = C.c_size_t
lib.gof_filter3d_ws_bytes.argtypes = [C.c_int64]
lib.gof_compute_3d_filter.restype = C.c_int
lib.gof_compute_3d_filter.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_int32), C.c_void_p]
CAM_FLOATS = 16


def camera_table(cameras, device):
    """(num_cams, 16) fp32: R (row-major, as the reference's torch.tensor(camera.R), :275), T (:276), focal_x, focal_y, W, H."""
    tab = np.zeros((len(cameras), CAM_FLOATS), dtype=np.float32)
    for i, cam in enumerate(cameras):
        tab[i, 0:9] = np.asarray(cam.R, dtype=np.float32).reshape(9)
        tab[i, 9:12] = np.asarray(cam.T, dtype=np.float32).reshape(3)
        tab[i, 12], tab[i, 13] = cam.focal_x, cam.focal_y
        tab[i, 14], tab[i, 15] = cam.image_width, cam.image_height
    return torch.from_numpy(tab).to(device)


def filter_3d(xyz, cam_table):
    """xyz (P,3) fp32 on a ROCm device, cam_table from camera_table() -> filter_3D (P,1)."""
    pts = B._need_cuda_f32(xyz, "xyz")
    tab = B._need_cuda_f32(cam_table, "camera table")
    n, ncam = int(pts.shape[0]), int(tab.shape[0])
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    if n == 0:
        return out[..., None]
    nb = lib.gof_filter3d_ws_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=pts.device)
    flag = C.c_int32(0)
    with torch.cuda.device(pts.device):
        B._check(lib.gof_compute_3d_filter(n, pts.data_ptr(), ncam, tab.data_ptr() if ncam else None, out.data_ptr(), ws.data_ptr(), nb,
                                           C.byref(flag), B._stream()))
    if not flag.value:
        # the reference fails here too: `distance[valid_points].max()` of an empty tensor (gaussian_model.py:306)
        raise RuntimeError("max(): Expected reduction dim to be specified for input.numel() == 0. (compute_3D_filter: no Gaussian is seen by any camera)")
    return out[..., None]


# A view-sharded trainer (launch/run_train_dp.py) hands train.py only this rank's cameras; the 3D filter must still be computed
# from ALL training cameras or the replicas diverge.  The launcher installs a function here that maps the list
# train.py passes to the full list.
CAMERA_LIST_HOOK = None


@torch.no_grad()
def compute_3D_filter(self, cameras):
    """Method replacement for GaussianModel.compute_3D_filter(self, cameras)."""
    print("Computing 3D filter")
    xyz = self.get_xyz
    if CAMERA_LIST_HOOK is not None:
        cameras = CAMERA_LIST_HOOK(cameras)
    cams = list(cameras)
    key = (id(cameras), len(cams), str(xyz.device))
    cached = getattr(self, "_gof_cam_table", None)
    if cached is None or cached[0] != key:
        cached = (key, camera_table(cams, xyz.device))
        self._gof_cam_table = cached
    self.filter_3D = filter_3d(xyz.detach(), cached[1])


lib.gof_add_densification_stats.restype = C.c_int
lib.gof_add_densification_stats.argtypes = [C.c_int64] + [C.c_void_p] * 7


@torch.no_grad()
def add_densification_stats(self, viewspace_point_tensor, update_filter):
    """Method replacement for GaussianModel.add_densification_stats (scene/gaussian_model.py:709-714): one launch instead of four
    boolean-mask read-modify-writes (each with a host sync).  The accumulators are updated in place."""
    grad = viewspace_point_tensor.grad
    if grad is None:
        raise AttributeError("'NoneType' object is not subscriptable (viewspace_point_tensor has no gradient)")
    n = int(grad.shape[0])
    g = B._need_cuda_f32(grad, "viewspace_point_tensor.grad")
    if update_filter.dtype != torch.bool or update_filter.shape[0] != n:
        raise IndexError("add_densification_stats: update_filter must be a bool mask over the %d Gaussians" % n)
    f = update_filter.contiguous().view(torch.uint8)
    bufs = []
    for name in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom"):
        t = getattr(self, name)
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or t.device != g.device:
            raise RuntimeError("add_densification_stats: %s must be a contiguous float32 (%d,1) tensor on %s" % (name, n, g.device))
        bufs.append(t)
    with torch.cuda.device(g.device):
        B._check(lib.gof_add_densification_stats(n, g.data_ptr(), f.data_ptr(), *[t.data_ptr() for t in bufs], B._stream()))
