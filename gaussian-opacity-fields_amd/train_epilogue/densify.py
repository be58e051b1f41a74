"""Drop-in for ``GaussianModel.densify_and_prune`` (reference scene/gaussian_model.py:685-707, with densify_and_clone :658-683,
densify_and_split :631-656, densification_postfix :609-629, prune_points :569-585 and the optimizer surgery :532-607 folded in).

The reference expresses every step with boolean-mask indexing -- each ``x[mask]`` is a nonzero + gather with a host sync -- and
re-concatenates / re-indexes the six parameter tensors and their Adam moments four times per densification.  Here the decisions
become ordered index lists on the device (``gof_densify_select``), the new Gaussians' values are computed by the SAME torch
operations as the reference on the compact selections (same inputs, same order -> same values, and ``torch.normal`` consumes the
generator exactly as the reference does: first the clones' draw, then the splits'), and every tensor is rebuilt ONCE by a row
gather (``gof_rows_gather``).  Two small read-backs remain (list lengths, final count).  Result: the same Gaussians in the same
order as the reference produces.  launch/run_reference_script.py rebinds the method on the reference's class."""
import ctypes as C

import torch
from torch import nn

from . import _backend as B

lib = B.lib
_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float
lib.gof_densify_ws_bytes.restype = C.c_size_t
lib.gof_densify_ws_bytes.argtypes = [_i64]
lib.gof_densify_select.restype = C.c_int
lib.gof_densify_select.argtypes = [_i64, _vp, _vp, _vp, _vp, _f32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.POINTER(_i64), _vp]
lib.gof_compact_rows.restype = C.c_int
lib.gof_compact_rows.argtypes = [_i64, _vp, _vp, _vp, _vp, C.c_size_t, C.POINTER(_i64), _vp]
lib.gof_rows_gather.restype = C.c_int
lib.gof_rows_gather.argtypes = [_i64, _i32, _vp, _vp, _vp, _vp, _vp]

SKIP_GROUPS = ("appearance_embeddings", "appearance_network")          # gaussian_model.py:535, 551, 590: not per-Gaussian


def select(accum, accum_abs, denom, scale_max, max_grad, q_abs, size_threshold):
    """-> (role u8 [P], keep_idx, clone_idx, split_idx) with the index lists trimmed to their lengths (int32, ascending)."""
    P = int(accum.shape[0])
    dev = accum.device
    a, aa, d = (B._need_cuda_f32(t.reshape(-1), n) for t, n in ((accum, "xyz_gradient_accum"), (accum_abs, "xyz_gradient_accum_abs"), (denom, "denom")))
    sm = B._need_cuda_f32(scale_max.reshape(-1), "scale_max")
    q = B._need_cuda_f32(q_abs.reshape(-1), "Q")
    role = torch.empty(P, dtype=torch.uint8, device=dev)
    lists = [torch.empty(max(P, 1), dtype=torch.int32, device=dev) for _ in range(3)]
    nb = lib.gof_densify_ws_bytes(P)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    counts = (_i64 * 3)()
    with torch.cuda.device(dev):
        B._check(lib.gof_densify_select(P, a.data_ptr(), aa.data_ptr(), d.data_ptr(), sm.data_ptr(), float(max_grad), q.data_ptr(), float(size_threshold),
                                        role.data_ptr(), lists[0].data_ptr(), lists[1].data_ptr(), lists[2].data_ptr(), ws.data_ptr(), nb, counts, B._stream()))
    return (role,) + tuple(l[:int(c)] for l, c in zip(lists, counts))


def compact_rows(keep, src_rows=None):
    """Rows (or src_rows[row]) whose `keep` byte is non-zero, in order (int32)."""
    n = int(keep.shape[0])
    dev = keep.device
    k = keep.to(torch.uint8).contiguous()
    out = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    nb = lib.gof_densify_ws_bytes(n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    cnt = _i64(0)
    with torch.cuda.device(dev):
        B._check(lib.gof_compact_rows(n, k.data_ptr(), src_rows.data_ptr() if src_rows is not None else None, out.data_ptr(), ws.data_ptr(), nb,
                                      C.byref(cnt), B._stream()))
    return out[:int(cnt.value)]


def rows_gather(rows, src, extra):
    """out[r] = src[rows[r]] if rows[r] >= 0 else extra[-rows[r] - 1] (zeros if extra is None); trailing dimensions as src."""
    n = int(rows.shape[0])
    s = B._need_cuda_f32(src, "source tensor")
    per = int(s[0].numel()) if s.shape[0] else int(torch.tensor(s.shape[1:]).prod())
    out = torch.empty((n,) + tuple(s.shape[1:]), dtype=torch.float32, device=s.device)
    if n == 0:
        return out
    e = B._need_cuda_f32(extra, "new rows") if extra is not None and extra.shape[0] else None
    with torch.cuda.device(s.device):
        B._check(lib.gof_rows_gather(n, per, rows.data_ptr(), s.data_ptr(), e.data_ptr() if e is not None else None, out.data_ptr(), B._stream()))
    return out


@torch.no_grad()
def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
    """Method replacement for GaussianModel.densify_and_prune (same arguments, same return triple)."""
    from utils.general_utils import build_rotation                      # the reference's own helper (gaussian_model.py:16)
    N = 2
    n0 = int(self._xyz.shape[0])
    grads = self.xyz_gradient_accum / self.denom                        # :686-690
    grads[grads.isnan()] = 0.0
    grads_abs = self.xyz_gradient_accum_abs / self.denom
    grads_abs[grads_abs.isnan()] = 0.0
    ratio = (torch.norm(grads, dim=-1) >= max_grad).float().mean()      # :691
    Q = torch.quantile(grads_abs.reshape(-1), 1 - ratio)                # :692 (stays a device scalar: no read-back)
    scale = self.get_scaling                                            # exp(_scaling), (n0, 3)
    smax = torch.max(scale, dim=1).values
    role, keep_idx, clone_idx, split_idx = select(self.xyz_gradient_accum, self.xyz_gradient_accum_abs, self.denom, smax, max_grad, Q,
                                                  self.percent_dense * extent)
    nc, ns = int(clone_idx.shape[0]), int(split_idx.shape[0])
    ci, si = clone_idx.long(), split_idx.long()

    # ---- values of the new Gaussians: the reference's statements on the compact selections, in the reference's order ----
    stds = scale.index_select(0, ci)                                    # densify_and_clone :668-672
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device="cuda"), std=stds)
    rots = build_rotation(self._rotation.index_select(0, ci))
    xyz_c = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz.index_select(0, ci)
    stds = scale.index_select(0, si).repeat(N, 1)                       # densify_and_split :645-652
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device="cuda"), std=stds)
    rots = build_rotation(self._rotation.index_select(0, si)).repeat(N, 1, 1)
    xyz_s = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz.index_select(0, si).repeat(N, 1)
    scaling_s = self.scaling_inverse_activation(scale.index_select(0, si).repeat(N, 1) / (0.8 * N))

    def new_rows(t, first=None, second=None):
        a = t.index_select(0, ci) if first is None else first
        b = t.index_select(0, si).repeat(*([N] + [1] * (t.dim() - 1))) if second is None else second
        return torch.cat((a, b), dim=0)
    extra = {"xyz": new_rows(self._xyz, xyz_c, xyz_s), "f_dc": new_rows(self._features_dc), "f_rest": new_rows(self._features_rest),
             "opacity": new_rows(self._opacity), "scaling": new_rows(self._scaling, None, scaling_s), "rotation": new_rows(self._rotation)}

    # ---- rows before the final prune: [originals that were not split, clones, split samples] (:609-629, :654-656) ----
    n_keep = int(keep_idx.shape[0])
    new_ids = -(torch.arange(nc + N * ns, device=keep_idx.device, dtype=torch.int32) + 1)
    rows = torch.cat((keep_idx, new_ids))
    # final prune (:699-704) on those rows; max_radii2D was zeroed by densification_postfix, which the reference has called by now
    opac = self.opacity_activation(torch.cat((self._opacity.index_select(0, keep_idx.long()), extra["opacity"]), dim=0))
    prune = (opac < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = torch.zeros_like(prune) if max_screen_size >= 0 else torch.ones_like(prune)      # zeros > max_screen_size
        smax_rows = torch.cat((smax.index_select(0, keep_idx.long()), smax.index_select(0, ci),
                               torch.max(self.scaling_activation(scaling_s), dim=1).values))
        prune = torch.logical_or(torch.logical_or(prune, big_vs), smax_rows > 0.1 * extent)
    final = compact_rows(torch.logical_not(prune), rows)
    n_final = int(final.shape[0])

    # ---- every per-Gaussian tensor and its Adam moments rebuilt once (:532-607) ----
    optimizable = {}
    for group in self.optimizer.param_groups:
        if group["name"] in SKIP_GROUPS:
            continue
        assert len(group["params"]) == 1
        old = group["params"][0]
        new = nn.Parameter(rows_gather(final, old.detach(), extra[group["name"]]).requires_grad_(True))
        state = self.optimizer.state.get(old, None)
        if state is not None:
            state["exp_avg"] = rows_gather(final, state["exp_avg"], None)
            state["exp_avg_sq"] = rows_gather(final, state["exp_avg_sq"], None)
            del self.optimizer.state[old]
            self.optimizer.state[new] = state
        group["params"][0] = new
        optimizable[group["name"]] = new
    self._xyz, self._features_dc, self._features_rest = optimizable["xyz"], optimizable["f_dc"], optimizable["f_rest"]
    self._opacity, self._scaling, self._rotation = optimizable["opacity"], optimizable["scaling"], optimizable["rotation"]
    for name in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom"):      # :625-629 (+ the prune of zeros)
        setattr(self, name, torch.zeros((n_final, 1), device="cuda"))
    self.max_radii2D = torch.zeros((n_final), device="cuda")
    before, clone, split = n0, n0 + nc, n0 + nc + N * ns - ns
    return clone - before, split - clone, split - n_final
