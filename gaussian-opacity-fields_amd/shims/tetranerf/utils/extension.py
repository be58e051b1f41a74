"""Import shim for ``tetranerf.utils.extension.cpp`` (reference submodules/tetra-triangulation, CPU CGAL
Delaunay; extract_mesh.py:51).  OUT OF SCOPE: CGAL is not in this image; SciPy's
Qhull Delaunay is the stand-in so that ``extract_mesh.py`` runs."""
import torch


class cpp:  # noqa: N801  (mirrors the reference's attribute access `cpp.triangulate`)
    @staticmethod
    def triangulate(points: torch.Tensor) -> torch.Tensor:
        from scipy.spatial import Delaunay
        tri = Delaunay(points.detach().cpu().double().numpy())
        return torch.from_numpy(tri.simplices.astype("int32")).to(points.device)
