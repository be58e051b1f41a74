"""Developer script: the REFERENCE's own kernels (oracle/_ref, hipcc build of the CUDA sources, default contraction)
timed on this GPU at S1M, next to the product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S, reference_binding as rb
from diff_gaussian_rasterization import _backend as B
sc = S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
ref = rb.Reference(sd, "")
dL = np.random.default_rng(1).normal(size=(9, sd["H"], sd["W"])).astype(np.float32)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ref.forward_nocopy() if hasattr(ref, "forward_nocopy") else ref.forward(); t1 = time.perf_counter()
    g = ref.backward(dL); t2 = time.perf_counter()
    print("reference (hipcc build) fwd %.2f ms (incl. D2H of image), bwd %.2f ms (incl. H2D/D2H)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), "R", ref.R)
