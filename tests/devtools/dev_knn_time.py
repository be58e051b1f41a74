"""Developer script: simple_knn.distCUDA2 -- product vs the REFERENCE's kernel (oracle/_ref) on this GPU."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
from simple_knn._C import distCUDA2
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libgof_knnref.so"))
L.knnref_mean_dist3.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for n in (100_000, 1_000_000, 5_000_000):
    p = torch.from_numpy(np.random.default_rng(0).uniform(-1.3, 1.3, (n, 3)).astype(np.float32)).cuda()
    out = torch.zeros(n, device="cuda")
    res = {}
    for name, fn in (("product", lambda: distCUDA2(p)), ("reference", lambda: L.knnref_mean_dist3(n, p.data_ptr(), out.data_ptr()))):
        fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); res[name] = (time.perf_counter() - t) * 1e3
    print("%d points: product %.2f ms, reference kernel %.2f ms" % (n, res["product"], res["reference"]))
