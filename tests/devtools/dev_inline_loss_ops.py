"""Developer script: what the INLINE torch code of the reference's loss (train.py:164-188: everything the launcher cannot rebind)
costs per iteration at 1600x1063, piece by piece (wall time of 50 repetitions, forward + backward where it applies)."""
import time, torch
dev = torch.device("cuda", 0)
H, W = 1063, 1600
rendering = torch.rand(9, H, W, device=dev, requires_grad=True)
depth_normal = torch.nn.functional.normalize(torch.rand(3, H, W, device=dev), dim=0)
wvt = torch.eye(4, device=dev); wvt[3, :3] = torch.tensor([0.1, 0.2, 0.3])
def timeit(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print("%-62s %.3f ms" % (name, 1e3 * (time.perf_counter() - t0) / n))
timeit("c2w = (world_view_transform.T).inverse()            (:177)", lambda: (wvt.T).inverse())
def normal_part():
    rn = torch.nn.functional.normalize(rendering[3:6], p=2, dim=0)
    c2w = (wvt.T).inverse()
    world = (c2w[:3, :3] @ rn.reshape(3, -1)).reshape(3, H, W)
    loss = (1 - (world * depth_normal).sum(dim=0)).mean()
    loss.backward()
timeit("normalize + c2w @ normal + error.mean(), fwd + bwd   (:174-182)", normal_part)
def dist_part():
    rendering[8].mean().backward()
timeit("distortion_map.mean(), fwd + bwd                      (:164-167)", dist_part)
gt = torch.rand(3, H, W, device=dev)
def l1_part():
    torch.abs(rendering[:3] - gt).mean().backward()
timeit("torch l1_loss(image, gt), fwd + bwd                   (:156)", l1_part)
def grad_accum():
    (rendering[:3].mean() + rendering[8].mean() + rendering[3:6].mean() + rendering[6].mean()).backward()
timeit("four slices of `rendering` meeting in one gradient (autograd's slice-backward adds)", grad_accum)
