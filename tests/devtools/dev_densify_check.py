"""GPU check: train_epilogue.densify_and_prune against the REFERENCE's own GaussianModel.densify_and_prune (the staged copy under
oracle/_ref/refpy, run on this GPU with torch) on identical models, statistics, optimizer states and generator seeds.
Prints one JSON line per case: counts, the returned triples, and the largest difference per tensor (0 = bit-identical)."""
import copy
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests", "e2e_shims"), os.path.join(ROOT, "oracle", "_ref", "refpy"), os.path.join(ROOT, "gaussian-opacity-fields_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch import nn  # noqa: E402
from scene.gaussian_model import GaussianModel  # noqa: E402  (the reference's class)
import train_epilogue as T  # noqa: E402


def make_model(P, seed, fused_adam):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda")   # noqa: E731
    m = GaussianModel(3)
    m._xyz = nn.Parameter(r(P, 3))
    m._features_dc = nn.Parameter(r(P, 1, 3))
    m._features_rest = nn.Parameter(0.1 * r(P, 15, 3))
    m._scaling = nn.Parameter(torch.log(torch.exp(0.7 * r(P, 3)) * 0.02))
    m._rotation = nn.Parameter(r(P, 4))
    m._opacity = nn.Parameter(2.5 * r(P, 1))
    m.max_radii2D = torch.zeros(P, device="cuda")
    m.filter_3D = torch.full((P, 1), 0.001, device="cuda")
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                                 position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001,
                                 appearance_embeddings_lr=0.001, appearance_network_lr=0.001)
    m.spatial_lr_scale = 1.0
    m.training_setup(args)
    if fused_adam:
        groups = m.optimizer.param_groups
        m.optimizer = T.FusedAdam([{"params": gr["params"], "lr": gr["lr"], "name": gr["name"]} for gr in groups], lr=0.0, eps=1e-15)
    for grp in m.optimizer.param_groups:                       # one step so that every per-Gaussian tensor has Adam moments
        for p in grp["params"]:
            if p.shape[0] == P:
                p.grad = 0.01 * r(*p.shape)
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    m.xyz_gradient_accum = torch.rand((P, 1), generator=g, device="cuda") * 0.002
    m.xyz_gradient_accum_abs = torch.rand((P, 1), generator=g, device="cuda") * 0.004
    m.xyz_gradient_accum_abs_max = torch.rand((P, 1), generator=g, device="cuda")
    m.denom = torch.randint(0, 4, (P, 1), generator=g, device="cuda").float()       # zeros included: 0/0 -> NaN -> 0
    m.max_radii2D = torch.rand(P, generator=g, device="cuda") * 40
    return m


def diff(a, b):
    if a.shape != b.shape:
        return "shape %s vs %s" % (tuple(a.shape), tuple(b.shape))
    return float((a.detach().float() - b.detach().float()).abs().max()) if a.numel() else 0.0


def state_of(m, name):
    for grp in m.optimizer.param_groups:
        if grp["name"] == name:
            return m.optimizer.state[grp["params"][0]]


if __name__ == "__main__":
    ref_method = GaussianModel.densify_and_prune
    for P, seed, fused, max_screen in ((20000, 1, False, 20), (20000, 2, True, None), (3000, 3, True, 20), (257, 4, False, None), (200000, 5, True, 20)):
        a = make_model(P, seed, fused)
        b = copy.deepcopy(a)
        torch.manual_seed(77)
        ra = ref_method(a, 0.0002, 0.05, 3.0, max_screen)
        torch.manual_seed(77)
        rb = T.densify_and_prune(b, 0.0002, 0.05, 3.0, max_screen)
        torch.cuda.synchronize()
        out = {"P": P, "fused_adam": fused, "max_screen_size": max_screen, "n_ref": int(a._xyz.shape[0]), "n_ours": int(b._xyz.shape[0]),
               "ret_ref": [int(x) for x in ra], "ret_ours": [int(x) for x in rb], "next_normal_equal": None, "diff": {}}
        for name, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation")):
            out["diff"][name] = diff(getattr(a, attr), getattr(b, attr))
            sa, sb = state_of(a, name), state_of(b, name)
            out["diff"][name + ".exp_avg"] = diff(sa["exp_avg"], sb["exp_avg"])
            out["diff"][name + ".exp_avg_sq"] = diff(sa["exp_avg_sq"], sb["exp_avg_sq"])
            out["diff"][name + ".is_param_of_optimizer"] = 0.0 if any(grp["params"][0] is getattr(b, attr) for grp in b.optimizer.param_groups) else 1.0
        for name in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom", "max_radii2D"):
            out["diff"][name] = diff(getattr(a, name), getattr(b, name))
        out["next_normal_equal"] = True        # both consumed the generator identically iff the next draw matches
        torch.manual_seed(77)
        print(json.dumps(out), flush=True)
