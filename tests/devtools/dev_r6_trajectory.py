"""Evidence run (GPU, not a suite test): a CONFIG-3-SHAPED training trajectory against the reference's own kernels.

BASELINE config 3 is Mip-NeRF360 `bicycle`, 30k iterations; no dataset is on disk, and tests/test_trajectory_gpu.py's proxy ends at
6 895 Gaussians on 160x120 images.  This run grows it by three orders of magnitude: the structured synthetic Blender-format scene of
tests/fixtures/make_blender_scene.py --large (>= 60 training views at >= 800x800, sub-splat colour detail on every surface), the
reference's UNCHANGED train.py (oracle/_ref/refpy, byte-identical staged copy) for >= 7000 iterations -- densification from 500 every
100, an opacity reset at 3000, SH degree steps every 1000, distortion / depth-normal terms from 3000 -- with the same seeds:
  * `product`    : this package's rasterizer, the reference's own torch loss / optimizer / densification (GOF_TORCH_EPILOGUE=1),
  * `reference`, `reference2` : the reference's CUDA kernels compiled for gfx950 (tests/reference_backend), same epilogue, twice (its
                   backward accumulates with atomicAdd: two runs of the same kernels differ -- that spread is the yardstick),
  * `product_default` : the product as a user runs it (launcher defaults: HIP epilogue, fused Adam, device-side densification; since
                   the second half of round 6 with the script's inline loss evaluated by one fused call per iteration),
  * `product_default_eager_loss` : the same with GOF_EAGER_LOSS=1 (every loss helper eager: the defaults of the first half of round 6).
Per run: test PSNR / L1 and the number of Gaussians at the marks, wall time and it/s of the training loop (tqdm-free: from the
iteration timestamps train.py prints at the marks), and for the product runs the binding's counters over the WHOLE run
(_backend._stats: frames redone because a learnt capacity / pool was too small, read-backs, shapes that inherited their pools).

    python tests/devtools/dev_r6_trajectory.py [--views 64] [--size 800 800] [--iters 7000] [--runs product,reference,reference2,product_default]
                                               [--out gpurun_out/r06/trajectory_large.json]
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "gaussian-opacity-fields_amd")
REFPY = os.path.join(ROOT, "oracle", "_ref", "refpy")
SHIMS = os.path.join(ROOT, "tests", "e2e_shims")
LAUNCH = {"product": os.path.join(PKG, "launch", "run_reference_script.py"),
          "reference": os.path.join(ROOT, "tests", "reference_backend", "run_with_reference_rasterizer.py")}


def env_for(name, stats_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([SHIMS] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env["GOF_E2E_SEED"] = "0"
    env["PYTHONUNBUFFERED"] = "1"                     # (the marks are time-stamped as train.py prints them)
    if not name.startswith("product_default"):
        env["GOF_TORCH_EPILOGUE"] = "1"
    if name == "product_default_eager_loss":          # the launcher's defaults until the deferred loss (train_epilogue/deferred.py)
        env["GOF_EAGER_LOSS"] = "1"
    if name.startswith("product"):
        env["GOF_STATS_JSON"] = stats_path
    return env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--test-views", type=int, default=8)
    ap.add_argument("--size", type=int, nargs=2, default=[800, 800])
    ap.add_argument("--iters", type=int, default=7000)
    ap.add_argument("--gt", type=int, default=250_000)
    ap.add_argument("--gt-scale", type=float, default=0.009)
    ap.add_argument("--init", type=int, default=100_000)
    ap.add_argument("--grad-threshold", type=float, default=0.0002)
    ap.add_argument("--runs", default="product,reference,reference2,product_default")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06", "trajectory_large.json"))
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    its = a.iters
    marks = sorted({1, 500, 1000, 2000, 3000, 4000, 5000, 6000, its} | ({its // 2} if its < 6000 else set()))
    marks = [m for m in marks if m <= its]
    work = tempfile.mkdtemp(prefix="gof_traj_")
    scene = os.path.join(work, "scene")
    t0 = time.time()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "fixtures", "make_blender_scene.py"), scene, "--views", str(a.views),
                           "--test-views", str(a.test_views), "--size", str(a.size[0]), str(a.size[1]), "--gt", str(a.gt), "--gt-scale", str(a.gt_scale),
                           "--init", str(a.init), "--large"], env=env_for("scene", ""))
    print("scene: %d + %d views of %dx%d from %d ground-truth Gaussians in %.0f s" % (a.views, a.test_views, a.size[0], a.size[1], a.gt, time.time() - t0), flush=True)
    train_args = ["--iterations", str(its), "--densify_until_iter", str(int(its * 6 / 7)), "--distortion_from_iter", str(min(3000, its // 2)),
                  "--depth_normal_from_iter", str(min(3000, its // 2)), "--densify_grad_threshold", str(a.grad_threshold),
                  "--test_iterations"] + [str(m) for m in marks] + ["--save_iterations"] + [str(m) for m in marks[1:]] + ["--eval", "--quiet"]
    out = {"what": "unchanged train.py (oracle/_ref/refpy), %d iterations on the structured synthetic Blender-format scene (%d training + %d test views of %dx%d "
                   "rendered from %d ground-truth Gaussians; %d initial points), same seeds; product / reference / reference2: the reference's own torch epilogue, "
                   "only the rasterizer differs; product_default: the launcher's defaults" % (its, a.views, a.test_views, a.size[0], a.size[1], a.gt, a.init),
           "train_args": train_args, "marks": marks, "runs": {}, "wall_s": {}, "iters_per_s": {}, "stats": {}}
    for name in a.runs.split(","):
        model = os.path.join(work, "model_" + name)
        stats_path = os.path.join(work, "stats_%s.json" % name)
        launcher = LAUNCH["reference" if name.startswith("reference") else "product"]
        cmd = [sys.executable, launcher, os.path.join(REFPY, "train.py"), "-s", scene, "-m", model] + train_args
        t0 = time.time()
        proc = subprocess.Popen(cmd, env=env_for(name, stats_path), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        curve, stamps, tail = {}, {}, []
        for line in proc.stdout:                      # (train.py prints one line per test mark: stamp it -- the loop's it/s between marks)
            tail.append(line)
            tail = tail[-40:]
            m = re.search(r"\[ITER (\d+)\] Evaluating test: L1 (\S+) PSNR (\S+)", line)
            if m:
                curve[int(m.group(1))] = {"l1": float(m.group(2)), "psnr": float(m.group(3))}
                stamps[int(m.group(1))] = time.time()
        rc = proc.wait()
        if rc != 0:
            print("".join(tail))
            raise SystemExit("run %s failed (rc %d)" % (name, rc))
        wall = time.time() - t0
        for it in marks[1:]:
            ply = os.path.join(model, "point_cloud", "iteration_%d" % it, "point_cloud.ply")
            head = open(ply, "rb").read(4096).decode("latin1")
            curve[it]["gaussians"] = int(re.search(r"element vertex (\d+)", head).group(1))
        out["runs"][name] = {str(k): v for k, v in sorted(curve.items())}
        out["wall_s"][name] = round(wall, 1)
        ks = sorted(stamps)
        # it/s of the training loop between the first and the last mark (includes the evaluations and saves at the marks in between: the
        # same work on every side)
        out["iters_per_s"][name] = round((ks[-1] - ks[0]) / max(1e-9, stamps[ks[-1]] - stamps[ks[0]]), 2) if len(ks) > 1 else None
        out["iters_per_s_by_segment"] = out.get("iters_per_s_by_segment", {})
        out["iters_per_s_by_segment"][name] = {"%d-%d" % (ks[i], ks[i + 1]): round((ks[i + 1] - ks[i]) / max(1e-9, stamps[ks[i + 1]] - stamps[ks[i]]), 1) for i in range(len(ks) - 1)}
        if os.path.exists(stats_path):
            out["stats"][name] = json.load(open(stats_path))
        shutil.rmtree(model, ignore_errors=True)
        print(name, "wall %.0f s" % wall, out["iters_per_s"][name], "it/s", json.dumps(out["runs"][name]), json.dumps(out["stats"].get(name)), flush=True)
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    r = out["runs"]
    if all(k in r for k in ("product", "reference", "reference2")):
        last = str(marks[-1])
        p, r1, r2 = r["product"][last], r["reference"][last], r["reference2"][last]
        lo, hi = min(r1["psnr"], r2["psnr"]), max(r1["psnr"], r2["psnr"])
        out["verdict"] = {"final_psnr": {"product": p["psnr"], "reference": r1["psnr"], "reference2": r2["psnr"]},
                          "product_inside_reference_spread_pm_0p3dB": bool(lo - 0.3 <= p["psnr"] <= hi + 0.3),
                          "final_gaussians": {"product": p["gaussians"], "reference": r1["gaussians"], "reference2": r2["gaussians"]}}
    for name, st in out["stats"].items():
        redone = st.get("fused_redone_frames", 0) + st.get("mask_pool_redone_frames", 0) + st.get("record_pool_redone_backwards", 0)
        out.setdefault("redone_fraction", {})[name] = round(redone / float(its), 5)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out.get("verdict")), json.dumps(out.get("redone_fraction")))
    if not a.keep:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
