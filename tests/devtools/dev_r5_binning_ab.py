"""Developer script (round 5, GPU): A/B of library build variants and environment switches in ONE process -- torch imported once, the
scenes generated once, every variant's library loaded side by side (an environment switch is read once per loaded library, so a
switched run gets its own copy of the .so under another name).  Per variant and scene: ms per fwd+bwd step through the autograd
surface (HIP events around N steps) and the per-stage HIP-event times of the library's profiler.

    python tests/devtools/dev_r5_binning_ab.py <label>[:<tag>][:ENV=V,ENV=V] ...     ('' tag = the shipped library)
"""
import os
import shutil
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpu_common import to_dev, settings_from  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B  # noqa: E402

PKG = os.path.dirname(os.path.dirname(os.path.abspath(B.__file__)))
# the instance capacity the binding learns is keyed by (device, P, W, H): S1M-clustered (24.8 M instances) raises it for S1M (8.8 M) as well --
# "S1M-overcap" is S1M again behind it (a capacity 3.5x the count: what a view with few instances pays for one with many); every variant
# starts from forgotten capacities
SCENES = (("S1M", lambda: S.scene_frustum(1_000_000, seed=0), 100),
          ("S1M-clustered", lambda: S.scene_clustered(1_000_000, seed=0), 30),
          ("S1M-overcap", None, 100),
          ("6M@1237x822", lambda: S.scene_frustum(6_000_000, W=1237, H=822, focal=1237.0 * 0.75, seed=0, sigma_px=1.5), 30))
only = os.environ.get("AB_SCENES")
if only:
    SCENES = tuple(s for s in SCENES if s[0] in only.split(","))


def load_variant(tag, env):
    path = os.path.join(PKG, "lib", "libgof_hip%s.so" % ("_" + tag if tag else ""))
    for k, v in env.items():
        os.environ[k] = v
    if env:                                     # a fresh copy: its load-time statics read THIS environment
        tmp = os.path.join(tempfile.mkdtemp(), "libgof_hip_%s_%s.so" % (tag or "shipped", "_".join(env)))
        shutil.copy(path, tmp)
        path = tmp
    B.LIB_PATH = path
    B.lib = B._load()
    return path


def main():
    specs = []
    for a in sys.argv[1:]:
        parts = a.split(":")
        label = parts[0]
        tag = parts[1] if len(parts) > 1 else ""
        env = dict(kv.split("=") for kv in parts[2].split(",")) if len(parts) > 2 and parts[2] else {}
        specs.append((label, tag, env))
    scenes = []
    for name, make, steps in SCENES:
        sd = to_dev(make()) if make else scenes[0][1]
        scenes.append((name, sd, steps))
    for label, tag, env in specs:
        load_variant(tag, env)
        for dct in (B._capacity, B._mask_need, B._staged_need):
            dct.clear()
        for name, sd, steps in scenes:
            params = {k: sd[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
            means2D = torch.zeros_like(params["means3D"], requires_grad=True)
            rast = GaussianRasterizer(settings_from(sd))
            dL = torch.randn((9, sd["H"], sd["W"]), device="cuda")

            def step():
                for p in params.values():
                    p.grad = None
                color, _ = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"], opacities=params["opacities"],
                                scales=params["scales"], rotations=params["rotations"])
                color.backward(dL)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(2):                   # two timed blocks, the faster one counts (the first absorbs the pools' learning)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / steps)
            B.profile_enable(True)
            for _ in range(6):
                step()
            torch.cuda.synchronize()
            rep = B.profile_report()
            B.profile_enable(False)
            k = {n: round(v["total_ms"] / v["calls"], 4) for n, v in rep.items()}
            print("%-22s %-14s %.4f ms/step  %s" % (label, name, best, k), flush=True)
            del params, means2D, rast, dL
            torch.cuda.empty_cache()
        for kk in env:
            os.environ.pop(kk, None)


if __name__ == "__main__":
    main()
