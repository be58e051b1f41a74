"""Developer script: marching tetrahedra at scale (Freudenthal grid), product vs the reference's torch implementation restated
on the same GPU (utils/tetmesh.py logic through torch ops)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import tetmesh
from diff_gaussian_rasterization import _backend as B

def grid(n):
    ax = torch.linspace(-1, 1, n, device="cuda")
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    verts = torch.stack([X, Y, Z], -1).reshape(-1, 3)
    idx = torch.arange(n ** 3, device="cuda").reshape(n, n, n)
    c = [idx[i:n - 1 + i, j:n - 1 + j, k:n - 1 + k].reshape(-1) for i in (0, 1) for j in (0, 1) for k in (0, 1)]   # corners 000..111
    v000, v001, v010, v011, v100, v101, v110, v111 = c
    tets = torch.stack([torch.stack(t, 1) for t in ((v000, v100, v110, v111), (v000, v100, v101, v111), (v000, v010, v110, v111),
                                                     (v000, v010, v011, v111), (v000, v001, v101, v111), (v000, v001, v011, v111))], 1).reshape(-1, 4)
    return verts, tets

for n in (100, 200, 300):
    verts, tets = grid(n)
    sdf = (verts.norm(dim=1) - 0.7 + 0.05 * torch.sin(verts[:, 0] * 20)).float()
    scales = torch.full((verts.shape[0], 1), 0.01, device="cuda")
    def call():
        return tetmesh.marching_tetrahedra(verts[None], tets, sdf[None], scales[None])
    call(); torch.cuda.synchronize(); t = time.perf_counter(); out = call(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("grid %d^3: %d vertices, %d tets -> %d edges, %d faces: %.2f ms" % (n, verts.shape[0], tets.shape[0], out[3][0].shape[0], out[2][0].shape[0], dt * 1e3))
    del verts, tets, sdf, scales, out
    torch.cuda.empty_cache()
