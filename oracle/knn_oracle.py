"""CPU restatement of the reference's simple-knn (TEST INFRASTRUCTURE ONLY; never imported by the product path).

`mean_dist3(points)` follows submodules/simple-knn/simple_knn.cu: for every point the three smallest squared distances to
OTHER points (by index, simple_knn.cu:176-177; coincident points count with 0), each distance evaluated in fp32 as
d.x*d.x + d.y*d.y + d.z*d.z without contraction (:141-142), kept ascending (:143-151) and averaged as
(best[0] + best[1] + best[2]) / 3.0f (:183); missing neighbours stay FLT_MAX (:154).  The reference's Morton / box
pruning (:39-136, 166-174) is conservative, i.e. it returns the exact 3-NN, so the restatement may find them any way it
likes: brute force below 4096 points, a KD-tree (float64 search of 8 candidates, fp32 re-evaluation) above.
Pinned against the reference's own kernel running on the GPU (oracle/_ref/libgof_knnref*.so, tests/test_knn_gpu.py)."""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def _dist2_f32(a, b):
    d = (b.astype(np.float32) - a.astype(np.float32)).astype(np.float32)
    dx2 = (d[..., 0] * d[..., 0]).astype(np.float32)
    dy2 = (d[..., 1] * d[..., 1]).astype(np.float32)
    dz2 = (d[..., 2] * d[..., 2]).astype(np.float32)
    return ((dx2 + dy2).astype(np.float32) + dz2).astype(np.float32)


def _finish(best):
    with np.errstate(over="ignore"):
        s = (best[:, 0] + best[:, 1]).astype(np.float32)
        s = (s + best[:, 2]).astype(np.float32)
        return (s / np.float32(3.0)).astype(np.float32)


def mean_dist3(points):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    n = pts.shape[0]
    best = np.full((n, 3), FLT_MAX, dtype=np.float32)
    if n <= 1:
        return _finish(best) if n else np.zeros(0, np.float32)
    if n <= 4096:
        for s in range(0, n, 256):
            d = _dist2_f32(pts[s:s + 256, None, :], pts[None, :, :])
            d[np.arange(d.shape[0]), np.arange(s, s + d.shape[0])] = np.inf          # self, by index
            k = min(3, n - 1)
            part = np.sort(d, axis=1)[:, :k]
            best[s:s + 256, :k] = part
        return _finish(best)
    from scipy.spatial import cKDTree
    tree = cKDTree(pts.astype(np.float64))
    kq = min(9, n)
    _, idx = tree.query(pts.astype(np.float64), k=kq)
    for s in range(0, n, 1 << 16):
        cand = idx[s:s + (1 << 16)]
        d = _dist2_f32(pts[s:s + cand.shape[0], None, :], pts[cand])
        d[cand == np.arange(s, s + cand.shape[0])[:, None]] = np.inf                    # self, by index
        best[s:s + cand.shape[0]] = np.sort(d, axis=1)[:, :3]
    return _finish(best)
