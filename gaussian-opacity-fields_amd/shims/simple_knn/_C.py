"""Import shim for ``simple_knn._C`` (reference submodules/simple-knn, CUDA; called once at
initialisation, scene/gaussian_model.py:327).  OUT OF SCOPE for kernels:
this torch stand-in only keeps ``train.py`` importable/runnable without the CUDA extension."""
import torch


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance to the 3 nearest neighbours (submodules/simple-knn/spatial.cu:15-26)."""
    pts = points.detach().float()
    n = pts.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    chunk = max(1, min(n, (1 << 26) // max(n, 1)))
    for s in range(0, n, chunk):
        d2 = torch.cdist(pts[s:s + chunk], pts).square_()
        k = min(4, n)
        nn = torch.topk(d2, k, dim=1, largest=False).values[:, 1:]
        out[s:s + chunk] = nn.mean(dim=1) if nn.numel() else 0.0
    return out
