"""Developer script: time the HIP training epilogue against the reference's torch implementation on the GPU
(1600x1063 image, 1M Gaussians x 59 floats for Adam).  Prints a JSON dict of ms per call."""
import json, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpu_common import *   # noqa
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import train_epilogue_oracle as O
import train_epilogue as T

dev = "cuda:0"
W, H = 1600, 1063


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


g = torch.Generator().manual_seed(0)
img = torch.rand((3, H, W), generator=g).to(dev).requires_grad_(True)
gt = torch.rand((3, H, W), generator=g).to(dev)
depth = (3 + torch.rand((1, H, W), generator=g)).to(dev).requires_grad_(True)
wvt = torch.eye(4, device=dev)
view = types.SimpleNamespace(world_view_transform=wvt, image_width=W, image_height=H, FoVx=1.18, FoVy=0.83)
out = {}


def ssim_fb(fn):
    def run():
        img.grad = None
        fn(img, gt).backward()
    return run


class GpuOracleView:      # the reference's depth_to_normal restated on GPU tensors == what train.py runs today
    pass


def ref_d2n(d):
    import math
    c2w = (wvt.T).inverse()
    fx = W / (2 * math.tan(view.FoVx / 2.)); fy = H / (2 * math.tan(view.FoVy / 2.))
    intr = torch.tensor([[fx, 0., W / 2.], [0., fy, H / 2.], [0., 0., 1.0]]).float().to(dev)
    gx, gy = torch.meshgrid(torch.arange(W, device=dev).float() + 0.5, torch.arange(H, device=dev).float() + 0.5, indexing='xy')
    pts = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).reshape(-1, 3)
    rays = pts @ intr.inverse().T @ c2w[:3, :3].T
    P = (d.reshape(-1, 1) * rays + c2w[:3, 3]).reshape(H, W, 3)
    o = torch.zeros_like(P)
    dx = P[2:, 1:-1] - P[:-2, 1:-1]; dy = P[1:-1, 2:] - P[1:-1, :-2]
    o[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return o


wn = torch.randn((H, W, 3), generator=g).to(dev)


def d2n_fb(fn):
    def run():
        depth.grad = None
        (fn(depth) * wn).sum().backward()
    return run


out["ssim_fwd_bwd_hip"] = timeit(ssim_fb(T.ssim))
out["ssim_fwd_bwd_torch"] = timeit(ssim_fb(O.ssim))
out["depth_to_normal_fwd_bwd_hip"] = timeit(d2n_fb(lambda d: T.depth_to_normal(view, d)[0]))
out["depth_to_normal_fwd_bwd_torch"] = timeit(d2n_fb(ref_d2n))

P = 1_000_000
shapes = [(3,), (1, 3), (15, 3), (1,), (3,), (4,)]
for name, cls, kw in (("hip", T.FusedAdam, {}), ("torch_foreach", torch.optim.Adam, {}), ("torch_fused", torch.optim.Adam, {"fused": True})):
    ps = [torch.nn.Parameter(torch.randn((P,) + s, device=dev)) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p)
    opt = cls([{"params": [p], "lr": 1e-3, "name": str(i)} for i, p in enumerate(ps)], lr=0.0, eps=1e-15, **kw)
    out["adam_" + name] = timeit(opt.step)
T._backend.lib.gof_profile_enable(1)
opt = T.FusedAdam([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15)
for _ in range(10):
    opt.step(); ssim_fb(T.ssim)(); d2n_fb(lambda d: T.depth_to_normal(view, d)[0])()
import ctypes
buf = ctypes.create_string_buffer(1 << 16)
T._backend.lib.gof_profile_report(buf, len(buf))
prof = json.loads(buf.value.decode())
out["kernels_ms"] = {k: v["total_ms"] / v["calls"] for k, v in prof.items()}
out["adam_GBps"] = 28.0 * 59 * P / (out["kernels_ms"]["adam_step"] * 1e-3) / 1e9
print(json.dumps(out, indent=1))
