"""Soak: 300 full training iterations (one-call loss + SplitSH + FusedAdam) at S1M; memory must stay flat, the loss finite."""
import os, sys, math, types, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("gaussian-opacity-fields_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import synthetic_scenes as S
import train_epilogue as T
from gpu_common import to_dev, settings_from
from diff_gaussian_rasterization import GaussianRasterizer, SplitSH
dev = torch.device("cuda", 0)
P, W, H = 1_000_000, 1600, 1063
sd = to_dev(S.scene_frustum(P, W=W, H=H, focal=1200.0, seed=0), dev)
raw = {"xyz": sd["means3D"].clone(), "f_dc": sd["shs"][:, :1].clone(), "f_rest": sd["shs"][:, 1:].clone(),
       "opacity": torch.logit(sd["opacities"].clamp(1e-4, 1 - 1e-4)), "scaling": torch.log(sd["scales"]), "rotation": sd["rotations"].clone()}
lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3}
params = {k: torch.nn.Parameter(v.contiguous()) for k, v in raw.items()}
opt = T.FusedAdam([{"params": [p], "lr": lrs[k], "name": k} for k, p in params.items()], lr=0.0, eps=1e-15)
rast = GaussianRasterizer(settings_from(sd))
with torch.no_grad():
    gt = rast(means3D=sd["means3D"], means2D=None, shs=sd["shs"], opacities=sd["opacities"], scales=sd["scales"], rotations=sd["rotations"])[0][:3].clone()
view = types.SimpleNamespace(world_view_transform=sd["viewmatrix"], image_width=W, image_height=H, FoVx=2 * math.atan(sd["tanfovx"]), FoVy=2 * math.atan(sd["tanfovy"]))
filter_3D = (sd["scales"].min(dim=1, keepdim=True).values * 0.1).contiguous()
A = T.activations
mem, losses = [], []
t0 = time.perf_counter()
for it in range(300):
    means2D = torch.zeros_like(params["xyz"], requires_grad=True)
    rendering, radii = rast(means3D=params["xyz"], means2D=means2D, shs=SplitSH(params["f_dc"], params["f_rest"]),
                            opacities=A.opacity_with_3D_filter(params["opacity"], params["scaling"], filter_3D),
                            scales=A.scaling_with_3D_filter(params["scaling"], filter_3D), rotations=A.rotation(params["rotation"]))
    loss = T.training_loss(rendering, gt, view, 0.2, 0.05, 100.0).loss
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    if it % 50 == 49:
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() >> 20); losses.append(round(loss.item(), 6))
torch.cuda.synchronize()
print("300 iterations in %.2f s (%.2f ms/it); allocated MiB every 50 its: %s; loss: %s; peak reserved %d MiB"
      % (time.perf_counter() - t0, (time.perf_counter() - t0) / 300 * 1e3, mem, losses, torch.cuda.max_memory_reserved() >> 20))
assert max(mem) - min(mem) < 64 and all(math.isfinite(l) for l in losses)
