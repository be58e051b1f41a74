"""Developer script (round 6, GPU): INTERLEAVED A/B of library build variants in one process.  dev_r5_binning_ab.py times its variants
one after the other, and a step time drifts by +-0.6 % over a process (clocks, the allocator's state): differences below 1 % drown.
Here every scene is timed in ROUNDS -- round r runs `steps` steps of variant 1, then of variant 2, ... -- and a variant's figure is
the MINIMUM and the MEDIAN over its rounds; the per-stage HIP-event times of the library's profiler come from a last pass.

    python tests/devtools/dev_r6_ab.py <label>[:<tag>[:<ATTR>=<literal>,...]] ...   ('' tag = the shipped library; ATTR: attributes of the
    binding module set while the variant runs, e.g. FUSED_FORWARD=False; AB_SCENES, AB_ROUNDS select)
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpu_common import to_dev, settings_from  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B  # noqa: E402

PKG = os.path.dirname(os.path.dirname(os.path.abspath(B.__file__)))
SCENES = (("S1M", lambda: S.scene_frustum(1_000_000, seed=0), 60),
          ("S1M-clustered", lambda: S.scene_clustered(1_000_000, seed=0), 30),
          ("6M@1237x822", lambda: S.scene_frustum(6_000_000, W=1237, H=822, focal=1237.0 * 0.75, seed=0, sigma_px=1.5), 25))
only = os.environ.get("AB_SCENES")
if only:
    SCENES = tuple(s for s in SCENES if s[0] in only.split(","))
ROUNDS = int(os.environ.get("AB_ROUNDS", "5"))


def main():
    if len(sys.argv) - 1 > 3:
        print("WARNING: more than three libraries in one process -- one of them will run ~11 %% slow whatever its code (profiles/r06_ab_call9_scalar_prelude.txt); "
              "do not read the row whose preprocess_fwd takes 0.06 ms", file=sys.stderr)
    variants = []
    for a in sys.argv[1:]:
        parts = a.split(":")
        tag = parts[1] if len(parts) > 1 else ""
        path = os.path.join(PKG, "lib", "libgof_hip%s.so" % ("_" + tag if tag else ""))
        B.LIB_PATH = path
        import ast
        attrs = {kv.split("=")[0]: ast.literal_eval(kv.split("=")[1]) for kv in parts[2].split(",")} if len(parts) > 2 and parts[2] else {}
        variants.append((parts[0], (B._load(), attrs)))
    for name, make, steps in SCENES:
        sd = to_dev(make())
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
        times = {label: [] for label, _ in variants}
        defaults = {}

        def select(variant):
            lib, attrs = variant
            B.lib = lib
            for k, v in defaults.items():
                setattr(B, k, v)
            for k, v in attrs.items():
                defaults.setdefault(k, getattr(B, k))
                setattr(B, k, v)
        for label, lib in variants:                     # every variant learns its pools once (they share the binding's dictionaries: same sizes)
            select(lib)
            for _ in range(5):
                step()
        torch.cuda.synchronize()
        for _ in range(ROUNDS):
            for label, lib in variants:
                select(lib)
                step(); step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                times[label].append(e0.elapsed_time(e1) / steps)
        for label, lib in variants:
            select(lib)
            B.profile_enable(True)
            for _ in range(6):
                step()
            torch.cuda.synchronize()
            rep = B.profile_report()
            B.profile_enable(False)
            k = {n: round(v["total_ms"] / v["calls"], 4) for n, v in rep.items()}
            t = times[label]
            print("%-14s %-14s min %.4f  median %.4f  (%s)  %s" % (label, name, min(t), statistics.median(t), " ".join("%.3f" % x for x in t), k), flush=True)
        del params, means2D, rast, dL, sd
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
