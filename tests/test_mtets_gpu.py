"""GPU: marching tetrahedra (csrc/mtets.hip) against the golden output of the reference's
utils/tetmesh.py (tests/golden/ref_python_golden.npz) and against the oracle restatement -- integer
results, bit-exact."""
import os

import numpy as np
import pytest
import torch

import oracle_binding as ob
import synthetic_scenes as S

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_golden.npz"))


def run(verts, tets, sdf, scales):
    from tetmesh import marching_tetrahedra
    d = "cuda"
    v = torch.from_numpy(verts).to(d)[None]
    out = marching_tetrahedra(v, torch.from_numpy(tets).to(d), torch.from_numpy(sdf).to(d)[None], torch.from_numpy(scales).to(d)[None, :, None])
    (pos, esdf), esc, faces, ids = [o[0] for o in out]
    torch.cuda.synchronize()
    return ids.cpu().numpy(), pos.cpu().numpy(), esdf.cpu().numpy(), esc.cpu().numpy(), faces.cpu().numpy()


def test_matches_reference_golden():
    ids, pos, esdf, esc, faces = run(GOLD["mt_verts"], GOLD["mt_tets"], GOLD["mt_sdf"], GOLD["mt_scales"])
    assert np.array_equal(ids, GOLD["mt_edge_ids"])
    assert np.array_equal(faces, GOLD["mt_faces"])
    assert np.array_equal(pos, GOLD["mt_edge_pos"])
    assert np.array_equal(esdf, GOLD["mt_edge_sdf"])
    assert np.array_equal(esc, GOLD["mt_edge_scales"])


def test_docstring_example():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    ids, pos, esdf, esc, faces = run(v, np.array([[0, 1, 2, 3]], np.int64), np.array([-1, -1, 0.5, 0.5], np.float32), np.ones(4, np.float32))
    assert np.array_equal(faces, np.array([[3, 0, 1], [3, 2, 0]]))
    assert np.array_equal(ids, GOLD["mt_doc_edge_ids"])


@pytest.mark.parametrize("n", [(3, 3, 3), (24, 20, 16)])
def test_matches_oracle_on_grids_and_degenerate_fields(n):
    verts, tets = S.freudenthal_tets(*n)
    rng = np.random.default_rng(7)
    centre = np.array(n, np.float32) / 2
    sdf = (0.4 * min(n) - np.linalg.norm(verts - centre, axis=1) + rng.normal(0, 0.3, len(verts))).astype(np.float32)
    scales = rng.uniform(0.1, 1, len(verts)).astype(np.float32)
    got = run(verts, tets, sdf, scales)
    want = ob.marching_tets(verts, tets, sdf, scales)
    for a, b in zip(got, [want[0], want[1], want[2][..., None], want[3][..., None], want[4]]):
        assert np.array_equal(a, b)
    assert got[4].min() >= 0 and got[4].max() < len(got[0])
    # all-outside and all-inside fields: no surface
    for const in (-1.0, 1.0):
        ids, pos, esdf, esc, faces = run(verts, tets, np.full(len(verts), const, np.float32), scales)
        assert len(ids) == 0 and len(faces) == 0
    # sdf exactly 0 counts as outside (sdf > 0 test, tetmesh.py:98)
    z = sdf.copy(); z[::3] = 0.0
    got = run(verts, tets, z, scales); want = ob.marching_tets(verts, tets, z, scales)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[4], want[4])


def test_more_than_32Mi_tets_face_order_follows_the_references_chunking():
    """utils/tetmesh.py:55-95 processes more than 32 Mi tets in torch.chunk pieces, which fixes the ORDER of the faces (chunk by
    chunk, inside a chunk the 1-triangle tets first).  47M tets = 2 chunks whose boundary is not a multiple of the kernels' 4096-tet
    blocks; bit-exact against the oracle (which restates the chunk loop) -- edges, positions and every face."""
    n = (200, 199, 201)
    # 6 tets per cell of an n0 x n1 x n2 vertex grid, built on the GPU (the numpy helper takes 25 s at this size)
    ax = [torch.arange(k, device="cuda", dtype=torch.float32) for k in n]
    X, Y, Z = torch.meshgrid(*ax, indexing="ij")
    verts = torch.stack([X, Y, Z], -1).reshape(-1, 3).cpu().numpy()
    idx = torch.arange(n[0] * n[1] * n[2], device="cuda").reshape(n)
    c = [idx[i:n[0] - 1 + i, j:n[1] - 1 + j, k:n[2] - 1 + k].reshape(-1) for i in (0, 1) for j in (0, 1) for k in (0, 1)]
    v000, v001, v010, v011, v100, v101, v110, v111 = c
    tets = torch.stack([torch.stack(t, 1) for t in ((v000, v100, v110, v111), (v000, v100, v101, v111), (v000, v010, v110, v111),
                                                     (v000, v010, v011, v111), (v000, v001, v101, v111), (v000, v001, v011, v111))], 1)
    tets = tets.reshape(-1, 4).cpu().numpy()
    del X, Y, Z, idx, c
    torch.cuda.empty_cache()
    assert len(tets) > 32 * 1024 * 1024
    rng = np.random.default_rng(11)
    centre = np.array(n, np.float32) / 2
    sdf = (0.37 * min(n) - np.linalg.norm(verts - centre, axis=1) + rng.normal(0, 0.2, len(verts)).astype(np.float32)).astype(np.float32)
    scales = rng.uniform(0.1, 1, len(verts)).astype(np.float32)
    got = run(verts, tets, sdf, scales)
    want = ob.marching_tets(verts, tets, sdf, scales)
    assert len(want[4]) > 100_000
    for a, b in zip(got, [want[0], want[1], want[2][..., None], want[3][..., None], want[4]]):
        assert np.array_equal(a, b)
