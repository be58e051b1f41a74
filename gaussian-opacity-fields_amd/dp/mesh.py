"""View-sharded opacity-field evaluation for mesh extraction ("integrate / mesh extraction shards by view as
well, with a min all-reduce over alpha_integrated").

``evaluate_alpha`` is the reference's ``evaluage_alpha`` (extract_mesh.py:17-34) with the view loop split across ranks: every
rank integrates its shard ``views[rank::world]`` -- with the per-view cache of diff_gaussian_rasterization, every rank keeps
only ITS views' Gaussian-side state in HBM, so N GPUs cache N times more views -- and the per-point minimum over views is
completed by one ``all_reduce(MIN)`` of ``(PN,)`` floats per call.  With ``return_color`` the colour of the view that attains
the minimum is selected exactly as the serial loop does (first view in list order wins ties, :28-29)."""
import torch
import torch.distributed as dist


@torch.no_grad()
def evaluate_alpha(points, views, integrate_fn, return_color=False, group=None):
    """integrate_fn(points, view) -> dict with "alpha_integrated" (PN,) and, if return_color, "color_integrated" (PN,3).
    Returns alpha = 1 - min over ALL views (and the colour of the arg-min view), identical on every rank."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    dev = points.device
    final_alpha = torch.ones((points.shape[0]), dtype=torch.float32, device=dev)              # extract_mesh.py:18
    final_color = torch.ones((points.shape[0], 3), dtype=torch.float32, device=dev) if return_color else None
    best_view = torch.full((points.shape[0],), len(views), dtype=torch.int64, device=dev)    # global index of the arg-min view
    for vi in range(rank, len(views), world):
        ret = integrate_fn(points, views[vi])
        alpha_integrated = ret["alpha_integrated"]
        better = alpha_integrated < final_alpha                                               # strict: the first view wins ties (:28)
        if return_color:
            final_color = torch.where(better.reshape(-1, 1), ret["color_integrated"], final_color)
        best_view = torch.where(better, torch.full_like(best_view, vi), best_view)
        final_alpha = torch.min(final_alpha, alpha_integrated)                                # :29
    if distributed:
        local_alpha = final_alpha.clone()
        dist.all_reduce(final_alpha, op=dist.ReduceOp.MIN, group=group)
        if return_color:
            # the serial loop keeps the FIRST view (lowest list index) that attains the minimum; a point no view improved keeps
            # the initial colour 1 on every rank
            cand = torch.where(local_alpha == final_alpha, best_view, torch.full_like(best_view, len(views) + 1))
            winner = cand.clone()
            dist.all_reduce(winner, op=dist.ReduceOp.MIN, group=group)
            mine = (cand == winner) & (winner < len(views))
            contrib = torch.where(mine.reshape(-1, 1), final_color, torch.zeros_like(final_color))
            dist.all_reduce(contrib, op=dist.ReduceOp.SUM, group=group)
            final_color = torch.where((winner < len(views)).reshape(-1, 1), contrib, torch.ones_like(contrib))
    alpha = 1 - final_alpha                                                                   # :31
    return (alpha, final_color) if return_color else alpha
