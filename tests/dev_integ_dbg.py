import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S, oracle_binding as ob
from diff_gaussian_rasterization import GaussianRasterizer
sc = S.scene_frustum(3000, W=96, H=64, focal=70.0, seed=8, kernel_size=0.1)
pts = S.tetra_points(sc)
o = ob.OracleScene(sc); oc, oal, ocol, orad = o.integrate(pts)
sd = to_dev(sc); r = GaussianRasterizer(settings_from(sd))
color, alpha, colp, radii = r.integrate(points3D=torch.from_numpy(pts).cuda(), means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"], scales=sd["scales"], rotations=sd["rotations"])
c = color.cpu().numpy()
for ch in range(9):
    bad = np.argwhere(bits(c[ch]) != bits(oc[ch]))
    print(ch, len(bad), [(tuple(b), float(c[ch][tuple(b)]), float(oc[ch][tuple(b)])) for b in bad[:3]])
a = alpha.cpu().numpy(); print("alpha mismatches", (bits(a) != bits(oal)).sum(), np.abs(a - oal).max())
