"""Developer script: the opacity-field query alone (S1M or, with `s5m`, the config-5 shape) for PMC collection / kernel traces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer
big = len(sys.argv) > 1 and sys.argv[1] == "s5m"
sc = S.scene_frustum(5_000_000, seed=0, sigma_px=1.5) if big else S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
pts = torch.from_numpy(S.tetra_points(sc)[::(9 if big else 3)].copy()).cuda()
r = GaussianRasterizer(settings_from(sd))
for _ in range(2):
    r.integrate(points3D=pts, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"], scales=sd["scales"], rotations=sd["rotations"])
torch.cuda.synchronize()
