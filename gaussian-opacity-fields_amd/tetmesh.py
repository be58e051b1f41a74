"""Marching tetrahedra on the GPU: drop-in for ``utils.tetmesh.marching_tetrahedra`` of the reference
(utils/tetmesh.py:141-190, imported by name at extract_mesh.py:14).

Same signature and return structure::

    verts_list, scale_list, faces_list, ids_list = marching_tetrahedra(vertices[None], tets, sdf[None], scales[None])
    (end_points (E,2,3), end_sdf (E,2,1)) = verts_list[0];  end_scales (E,2,1) = scale_list[0]
    faces (F,3) int64 = faces_list[0];  edge vertex ids (E,2) int64 = ids_list[0]

The work is done by ``gof_mtets_classify`` / ``gof_mtets_count`` / ``gof_mtets_emit`` of libgof_hip.so (csrc/mtets.hip): crossing
edges unique and sorted by (min id, max id) exactly as ``torch.unique(dim=0)`` orders them, faces in
the reference's order (per 32 Mi-tet chunk: 1-triangle tets first, then 2-triangle tets).
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _backend as B

__all__ = ["marching_tetrahedra"]


def _unbatched_marching_tetrahedra(vertices, tets, sdf, scales):
    dev = vertices.device
    if dev.type != "cuda":
        raise RuntimeError("marching_tetrahedra (gfx950 backend) needs tensors on a ROCm device, got %s" % dev)
    V, Tt = int(vertices.shape[0]), int(tets.shape[0])
    verts = vertices.detach().to(torch.float32).contiguous()
    t64 = tets.detach().to(torch.int64).contiguous()
    s32 = sdf.detach().to(torch.float32).reshape(-1).contiguous()
    sc32 = scales.detach().to(torch.float32).reshape(-1).contiguous()
    if s32.numel() != V or sc32.numel() != V:
        raise RuntimeError("sdf / scales must have one value per vertex")
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # per-tet state: 1 byte per tet; everything else is sized for the tets the surface actually crosses
        tws = torch.empty(int(B.lib.gof_mtets_tet_ws_bytes(Tt)), dtype=torch.uint8, device=dev)
        nv = C.c_int64(0)
        B._check(B.lib.gof_mtets_classify(V, Tt, B._ptr(t64), B._ptr(s32), B._ptr(tws), tws.numel(), C.byref(nv), stream))
        ews = torch.empty(int(B.lib.gof_mtets_edge_ws_bytes(int(nv.value))), dtype=torch.uint8, device=dev)
        ne, nf = C.c_int64(0), C.c_int64(0)
        B._check(B.lib.gof_mtets_count(V, Tt, B._ptr(t64), B._ptr(s32), B._ptr(tws), tws.numel(), B._ptr(ews), ews.numel(),
                                       C.byref(ne), C.byref(nf), stream))
        E, F = int(ne.value), int(nf.value)
        ids = torch.empty((E, 2), dtype=torch.int64, device=dev)
        pos = torch.empty((E, 2, 3), dtype=torch.float32, device=dev)
        esdf = torch.empty((E, 2, 1), dtype=torch.float32, device=dev)
        esc = torch.empty((E, 2, 1), dtype=torch.float32, device=dev)
        faces = torch.empty((F, 3), dtype=torch.int64, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731  (zero-sized outputs still need a non-NULL pointer)
        if E or F:
            B._check(B.lib.gof_mtets_emit(V, Tt, B._ptr(t64), B._ptr(verts), B._ptr(s32), B._ptr(sc32), B._ptr(tws), tws.numel(), B._ptr(ews), ews.numel(), E, F,
                                          p(ids), p(pos), p(esdf), p(esc), p(faces), stream))
    return (pos, esdf), esc, faces, ids


def marching_tetrahedra(vertices, tets, sdf, scales):
    outs = [_unbatched_marching_tetrahedra(vertices[b], tets, sdf[b], scales[b]) for b in range(vertices.shape[0])]
    return list(zip(*outs))
