"""Developer script: a few fwd+bwd iterations at S1M for PMC collection."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import _backend as B
sc = S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
dL = torch.randn(9, sd["H"], sd["W"], device="cuda")
for _ in range(3):
    res = product_forward_raw(sd)
    a = res["args"]
    B.rasterize_gaussians_backward(a[0], a[1], res["radii"], a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                                   dL, a[17], a[18], a[19], res["geom"], res["R"], res["binning"], res["img"], False)
torch.cuda.synchronize()
# the opacity-field query on the same scene (9 points per Gaussian): integrate_pixels / bin_points / integrate_points get counters too
from diff_gaussian_rasterization import GaussianRasterizer
pts = torch.from_numpy(S.tetra_points(sc)).cuda()
r = GaussianRasterizer(settings_from(sd))
for _ in range(2):
    r.integrate(points3D=pts, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"], scales=sd["scales"], rotations=sd["rotations"])
torch.cuda.synchronize()
