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
