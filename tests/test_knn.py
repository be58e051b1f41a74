"""simple_knn.distCUDA2 on gfx950: CPU checks of the oracle and the host-side API; GPU parity of the HIP
implementation against the oracle and against the REFERENCE's own kernel compiled for this GPU (oracle/_ref)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import knn_oracle as KO   # noqa: E402


def _cloud(n, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1.3, 1.3, (n, 3)).astype(np.float32)                 # scene/dataset_readers.py:241-247
    if kind == "clustered":
        c = rng.normal(0, 2.0, (max(1, n // 500), 3))
        return (c[rng.integers(0, len(c), n)] + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
    if kind == "line":
        t = rng.uniform(0, 1, (n, 1))
        return (t * np.array([[1.0, 2.0, 3.0]]) + 5.0).astype(np.float32)
    if kind == "duplicates":
        base = rng.uniform(0, 1, (max(1, n // 4), 3)).astype(np.float32)
        return base[rng.integers(0, len(base), n)]
    raise ValueError(kind)


def test_oracle_brute_force_and_tree_paths_agree_and_match_a_float64_knn():
    pts = _cloud(6000, 0)
    a = KO.mean_dist3(pts)                                   # KD-tree path
    d = ((pts[:1500, None, :].astype(np.float64) - pts[None].astype(np.float64)) ** 2).sum(-1)
    d[np.arange(1500), np.arange(1500)] = np.inf
    ref = np.sort(d, axis=1)[:, :3].mean(1)
    np.testing.assert_allclose(a[:1500], ref, rtol=2e-6)
    b = KO.mean_dist3(pts[:3000])                            # brute-force path
    d = ((pts[:3000, None, :].astype(np.float64) - pts[None, :3000].astype(np.float64)) ** 2).sum(-1)
    d[np.arange(3000), np.arange(3000)] = np.inf
    np.testing.assert_allclose(b, np.sort(d, axis=1)[:, :3].mean(1), rtol=2e-6)


def test_oracle_edge_cases():
    assert KO.mean_dist3(np.zeros((0, 3), np.float32)).shape == (0,)
    one = KO.mean_dist3(np.zeros((1, 3), np.float32))
    assert np.isinf(one[0])                                  # (3 * FLT_MAX) overflows, as in the reference's fp32 sum
    two = KO.mean_dist3(np.array([[0, 0, 0], [1, 0, 0]], np.float32))
    assert np.isinf(two).all()
    four = KO.mean_dist3(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]], np.float32))
    assert four[0] == pytest.approx((1 + 4 + 9) / 3)
    dup = KO.mean_dist3(np.zeros((5, 3), np.float32))
    assert (dup == 0).all()


def test_host_api_fails_loudly():
    from simple_knn._C import distCUDA2, lib
    with pytest.raises(RuntimeError, match="ROCm device"):
        distCUDA2(torch.zeros(10, 3))
    with pytest.raises(RuntimeError, match="dimensions"):
        distCUDA2(torch.zeros(10, 2))
    assert lib.gof_knn_ws_bytes(1_000_000) > 1_000_000 * 32
    assert lib.gof_knn_mean_dist3(-1, None, None, None, 0, None) < 0
    assert lib.gof_knn_mean_dist3(0, None, None, None, 0, None) == 0


def _product(pts):
    from simple_knn._C import distCUDA2
    return distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,kind", [(1, "uniform"), (2, "uniform"), (3, "uniform"), (4, "uniform"), (255, "uniform"), (257, "uniform"),
                                    (5000, "uniform"), (100_000, "uniform"), (60_000, "clustered"), (30_000, "line"), (20_000, "duplicates")])
def test_matches_oracle(n, kind):
    pts = _cloud(n, n + len(kind), kind)
    got, ref = _product(pts), KO.mean_dist3(pts)
    assert got.shape == ref.shape
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    # the 3 neighbours are exact; the squared distances are three fp32 products and two adds: bit-equal without contraction
    assert np.array_equal(got[fin], ref[fin])


def _reference(pts, variant="_nofma"):
    path = os.path.join(ROOT, "oracle", "_ref", "libgof_knnref%s.so" % variant)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libgof_knnref%s.so not built (needs /root/reference at build time)" % variant)
    L = ctypes.CDLL(path)
    L.knnref_mean_dist3.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    p = torch.from_numpy(pts).cuda()
    out = torch.zeros(len(pts), device="cuda")
    torch.cuda.synchronize()
    assert L.knnref_mean_dist3(len(pts), p.data_ptr(), out.data_ptr()) == 0
    return out.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,kind", [(4, "uniform"), (5000, "uniform"), (100_000, "uniform"), (60_000, "clustered"), (20_000, "duplicates"),
                                    (1_000_000, "uniform")])
def test_matches_the_references_own_kernel_on_this_gpu(n, kind):
    """Pins the oracle AND the product to the reference's simple_knn.cu compiled for gfx950 (no-contraction build: bit-equal;
    default build: the compiler may fuse the three products, 1e-6)."""
    pts = _cloud(n, 7 + n, kind)
    got = _product(pts)
    ref = _reference(pts, "_nofma")
    assert np.array_equal(got, ref)
    if n <= 100_000:
        assert np.array_equal(KO.mean_dist3(pts), ref)
    np.testing.assert_allclose(got, _reference(pts, ""), rtol=1e-6, atol=1e-12)
