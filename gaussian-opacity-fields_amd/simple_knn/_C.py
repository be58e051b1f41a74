"""Drop-in for the reference's pybind11 module ``simple_knn._C`` (submodules/simple-knn/ext.cpp, spatial.cu:15-26) on top
of ``gof_knn_mean_dist3`` (include/gof_knn_hip.h).  No fallback: tensors must live on a ROCm device."""
import ctypes as C

import torch

from diff_gaussian_rasterization import _backend as _B

lib = _B.lib
lib.gof_knn_ws_bytes.restype = C.c_size_t
lib.gof_knn_ws_bytes.argtypes = [C.c_int64]
lib.gof_knn_mean_dist3.restype = C.c_int
lib.gof_knn_mean_dist3.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest neighbours -> (N,) float32 on points.device
    (spatial.cu:15-26: `means = torch.full({P}, 0.0)`; SimpleKNN::knn)."""
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 (gfx950 backend) needs the points on a ROCm device, got %s" % points.device)
    if points.dtype != torch.float32:
        raise RuntimeError("distCUDA2: expected a float32 tensor (the reference reads data<float>()), got %s" % points.dtype)
    pts = points.contiguous()
    n = int(pts.size(0))
    out = torch.zeros(n, dtype=torch.float32, device=pts.device)
    if n:
        with torch.cuda.device(pts.device):
            nb = lib.gof_knn_ws_bytes(n)
            ws = torch.empty(nb, dtype=torch.uint8, device=pts.device)
            _B._check(lib.gof_knn_mean_dist3(n, pts.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, _B._stream()))
    return out
