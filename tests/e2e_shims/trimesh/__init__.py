"""Minimal `trimesh` stand-in (test infrastructure, see ../README.md): the unit box of scene/gaussian_model.py:434 and the mesh
container + PLY export of extract_mesh.py:111-120."""
import numpy as np

from . import creation  # noqa: F401


class Trimesh:
    def __init__(self, vertices=None, faces=None, vertex_colors=None, process=False, **kw):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
        self.vertex_colors = None if vertex_colors is None else np.asarray(vertex_colors, dtype=np.uint8)

    def update_vertices(self, mask):
        mask = np.asarray(mask, dtype=bool)
        remap = np.cumsum(mask) - 1
        self.vertices = self.vertices[mask]
        if self.vertex_colors is not None:
            self.vertex_colors = self.vertex_colors[mask]
        self.faces = remap[self.faces]          # callers drop the faces that lost a vertex with update_faces (as extract_mesh.py does)

    def update_faces(self, mask):
        self.faces = self.faces[np.asarray(mask, dtype=bool)]

    def export(self, path):
        nv, nf = len(self.vertices), len(self.faces)
        with open(path, "wb") as f:
            hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % nv
            if self.vertex_colors is not None:
                hdr += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
            hdr += "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % nf
            f.write(hdr.encode())
            if self.vertex_colors is None:
                f.write(self.vertices.astype("<f4").tobytes())
            else:
                rec = np.empty(nv, dtype=[("p", "<f4", 3), ("c", "u1", 3)])
                rec["p"] = self.vertices
                rec["c"] = self.vertex_colors[:, :3]
                f.write(rec.tobytes())
            fr = np.empty(nf, dtype=[("n", "u1"), ("v", "<i4", 3)])
            fr["n"] = 3
            fr["v"] = self.faces
            f.write(fr.tobytes())
