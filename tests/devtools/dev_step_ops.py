"""Developer script: which torch ops (and how many GPU kernels each) one steady-state fwd+bwd step of the S1M bench issues besides the
library's own launches -- torch.profiler over 5 steps, the aten ops with a device kernel listed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import synthetic_scenes as S
from gpu_common import to_dev, settings_from
from diff_gaussian_rasterization import GaussianRasterizer

sd = to_dev(S.scene_frustum(1_000_000, seed=0), "cuda:0")
names = ("means3D", "shs", "opacities", "scales", "rotations")
params = {k: sd[k].clone().requires_grad_(True) for k in names}
means2D = torch.zeros_like(params["means3D"], requires_grad=True)
rast = GaussianRasterizer(settings_from(sd))
dL = None


def step():
    global dL
    for p in params.values():
        p.grad = None
    means2D.grad = None
    color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    if dL is None:
        dL = torch.randn_like(color)
    color.backward(dL)


for _ in range(5):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
for e in prof.key_averages():
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if dt > 0:
        print("%-70s calls %4d  device us/call %8.1f" % (e.key[:70], e.count, dt / e.count))
