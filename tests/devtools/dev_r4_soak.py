"""Developer script: soak of the optimistic pools -- S1M seen from 8 posed cameras in a random order for N iterations (default 400),
the gradients of every iteration compared BIT FOR BIT with that view's gradients computed once on worst-case workspaces (the backward
is bit-reproducible), starting from nothing learnt: the views come in an order that makes later frames need more than any earlier
one (pool redos must happen and must be invisible).  Prints the redo counters and the time per iteration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "gaussian-opacity-fields_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import synthetic_scenes as S
from gpu_common import to_dev, settings_from
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sc = S.scene_frustum(1_000_000, seed=0)
views = [to_dev(sc if v == 0 else S.other_view(sc, v), "cuda:0") for v in range(8)]
names = ("means3D", "shs", "opacities", "scales", "rotations")
params = {k: views[0][k].clone().requires_grad_(True) for k in names}
means2D = torch.zeros_like(params["means3D"], requires_grad=True)
rasts = [GaussianRasterizer(settings_from(v)) for v in views]
dL = torch.randn((9, views[0]["H"], views[0]["W"]), generator=torch.Generator().manual_seed(1)).to("cuda:0")


def step(v):
    for p in params.values():
        p.grad = None
    means2D.grad = None
    color, radii = rasts[v](means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dL)
    return color.detach(), [params[k].grad for k in names] + [means2D.grad]


B.FULL_MASK_POOL = B.FULL_BACKWARD_SCRATCH = True
want = []
for v in range(8):
    step(v)
    c, g = step(v)
    want.append((c.clone(), [x.clone() for x in g]))
B.FULL_MASK_POOL = B.FULL_BACKWARD_SCRATCH = False
B._capacity.clear(); B._mask_need.clear(); B._staged_need.clear()
for k in B._stats:
    B._stats[k] = 0
rng = np.random.default_rng(5)
order = [7, 6, 5, 4, 3, 2, 1, 0] + [int(x) for x in rng.integers(0, 8, N)]      # lightest view first: every new view needs more than all before
torch.cuda.synchronize()
t0 = time.perf_counter()
for it, v in enumerate(order):
    c, g = step(v)
    if it < 16 or it % 7 == 0:
        assert torch.equal(c, want[v][0]), "image of view %d changed at iteration %d" % (v, it)
        for a, b, n in zip(g, want[v][1], names + ("means2D",)):
            assert torch.equal(a, b), "gradient %s of view %d differs at iteration %d (max %g)" % (n, v, it, (a - b).abs().max().item())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / len(order) * 1e3
print("%d iterations over 8 views, %.3f ms each (with the comparisons), every checked image and gradient bit-identical to the worst-case workspaces'" % (len(order), dt))
print("stats", dict(B._stats))
print("learnt: capacity", dict(B._capacity), "mask need", dict(B._mask_need), "staged need", dict(B._staged_need))
free, total = torch.cuda.mem_get_info()
print("device memory in use %.2f GiB, torch reserved %.2f GiB" % ((total - free) / 2**30, torch.cuda.memory_reserved() / 2**30))
