"""Developer script: cost of the library's per-launch HIP events (gof_profile_enable) on the bench step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B
sc = S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
means2D = torch.zeros_like(params["means3D"], requires_grad=True)
rast = GaussianRasterizer(settings_from(sd))
dL = torch.randn((9, sd["H"], sd["W"]), device="cuda")
def step():
    for p in params.values(): p.grad = None
    means2D.grad = None
    color, radii = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dL)
def run(n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(5): step()
for rep in range(2):
    B.profile_enable(False); a = run()
    B.profile_enable(True); b = run(); B.profile_report(); B.profile_enable(False)
    print("ms/step without events %.4f  with events %.4f" % (a, b))
