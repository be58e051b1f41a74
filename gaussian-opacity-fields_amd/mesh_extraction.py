"""The view loop of the mesh extraction with its reduction on the device.

``evaluate_alpha`` is the reference's ``evaluage_alpha`` (extract_mesh.py:17-34): the opacity field at `points` is the minimum over
all training views of ``alpha_integrated``, optionally with the colour of the view that attains it.  The reference materialises
``ones`` / ``zeros`` outputs per view and combines them with ``torch.min`` / ``torch.where`` (four passes over N points per view);
here the point pass of ``integrate`` min-combines into the running buffers in its final store (``integrate_min_into`` ->
``gof_integrate_points_min``), with the Gaussian side of every view served from the per-view cache.  Same values, bit for bit
(``tests/test_mesh_extraction_gpu.py``).  ``launch/run_reference_script.py`` binds it in place of the script's own function.
"""
import torch

from diff_gaussian_rasterization import integrate_min_into


@torch.no_grad()
def evaluate_alpha(points, views, gaussians, pipeline, background, kernel_size, return_color=False, integrate=None, progress=True):
    if integrate is None:
        from gaussian_renderer import integrate
    pts = points.detach().to(device="cuda", dtype=torch.float32).contiguous()
    final_alpha = torch.ones((pts.shape[0],), dtype=torch.float32, device=pts.device)                  # extract_mesh.py:18
    final_color = torch.ones((pts.shape[0], 3), dtype=torch.float32, device=pts.device) if return_color else None   # :20
    it = views
    if progress:
        try:
            from tqdm import tqdm
            it = tqdm(views, desc="Rendering progress")                                                   # :23
        except ImportError:
            pass
    with integrate_min_into(final_alpha, final_color):
        for view in it:
            integrate(pts, view, gaussians, pipeline, background, kernel_size=kernel_size)                # :24-29, reduction fused
    alpha = 1 - final_alpha                                                                               # :31
    if return_color:
        return alpha, final_color
    return alpha
