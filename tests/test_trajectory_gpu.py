"""GPU: TRAINING-TRAJECTORY parity against the reference's own kernels (proxy for BASELINE config 3, "PSNR vs reference": no dataset
is on disk, so the comparison runs on the synthetic Blender-format fixture).

The reference's UNCHANGED train.py (oracle/_ref/refpy, byte-identical staged copy) is run for 1500 iterations -- through
densification (from 100, every 100), three opacity resets, the SH-degree step at 1000, the distortion / depth-normal terms from 600
-- three times with the same seeds and the reference's own torch loss / optimizer / densification code (GOF_TORCH_EPILOGUE=1) on
every side, so that ONLY the rasterizer differs:
  * `product`    : this package (launch/run_reference_script.py),
  * `reference`  : the reference's CUDA kernels compiled for gfx950 (tests/reference_backend, oracle/_ref/libgof_cudaref.so),
  * `reference2` : the same again -- the reference's backward accumulates with atomicAdd, so its own trajectory differs from run to
                   run; that spread is the yardstick (the product is deterministic).
Compared at iterations 1, 100, 250, 500, 750, 1000, 1250, 1500: test PSNR, test L1, number of Gaussians (saved point clouds).  The
curves are written to gpurun_out/trajectory_parity.json (copied to profiles/ by the round's evidence pass)."""
import json
import os
import re
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-opacity-fields_amd")
REFPY = os.path.join(ROOT, "oracle", "_ref", "refpy")
SHIMS = os.path.join(ROOT, "tests", "e2e_shims")
pytestmark = pytest.mark.gpu

ITERS = 1500
MARKS = [1, 100, 250, 500, 750, 1000, 1250, 1500]
TRAIN_ARGS = ["--iterations", str(ITERS), "--densify_from_iter", "100", "--densification_interval", "100", "--opacity_reset_interval", "300",
              "--densify_until_iter", "1200", "--distortion_from_iter", "600", "--depth_normal_from_iter", "600",
              "--test_iterations"] + [str(m) for m in MARKS] + ["--save_iterations"] + [str(m) for m in MARKS[1:]] + ["--eval", "--quiet"]


def _env(**extra):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([SHIMS] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env.update(GOF_TORCH_EPILOGUE="1", GOF_E2E_SEED="0", **extra)      # (the seed: tests/e2e_shims/sitecustomize.py -- train.py itself seeds nothing)
    return env


def _start(launcher, scene, model):
    cmd = [sys.executable, launcher, os.path.join(REFPY, "train.py"), "-s", scene, "-m", model] + TRAIN_ARGS
    return cmd, time.time(), subprocess.Popen(cmd, env=_env(), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _finish(started, model):
    cmd, t0, proc = started
    stdout, stderr = proc.communicate(timeout=900)
    assert proc.returncode == 0, "command failed: %s\n--- stdout\n%s\n--- stderr\n%s" % (" ".join(cmd), stdout[-3000:], stderr[-5000:])
    curve = {}
    for m in re.finditer(r"\[ITER (\d+)\] Evaluating test: L1 (\S+) PSNR (\S+)", stdout):
        curve[int(m.group(1))] = {"l1": float(m.group(2)), "psnr": float(m.group(3))}
    for it in MARKS[1:]:
        ply = os.path.join(model, "point_cloud", "iteration_%d" % it, "point_cloud.ply")
        head = open(ply, "rb").read(4096).decode("latin1")
        curve[it]["gaussians"] = int(re.search(r"element vertex (\d+)", head).group(1))
    return curve, time.time() - t0


def test_training_trajectory_matches_the_references_own_kernels(tmp_path_factory):
    assert os.path.exists(os.path.join(REFPY, "train.py")), "oracle/_ref/refpy is missing: run __graft_entry__.build() where /root/reference exists"
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgof_cudaref.so")), "oracle/_ref/libgof_cudaref.so is missing"
    scene = str(tmp_path_factory.mktemp("blender_scene"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "fixtures", "make_blender_scene.py"), scene], env=_env())
    runs, secs, started = {}, {}, {}
    # the three runs side by side on the one GPU (a 160x120 scene is launch-latency bound: they overlap almost perfectly)
    for name, launcher in (("product", os.path.join(PKG, "launch", "run_reference_script.py")),
                           ("reference", os.path.join(ROOT, "tests", "reference_backend", "run_with_reference_rasterizer.py")),
                           ("reference2", os.path.join(ROOT, "tests", "reference_backend", "run_with_reference_rasterizer.py"))):
        model = str(tmp_path_factory.mktemp("model_" + name))
        started[name] = (_start(launcher, scene, model), model)
    for name, (st, model) in started.items():
        runs[name], secs[name] = _finish(st, model)
        assert sorted(runs[name]) == MARKS, (name, sorted(runs[name]))
    out = {"what": "unchanged train.py, %d iterations on the synthetic Blender fixture (24 views of 160x120), same seeds, the reference's own torch "
                   "epilogue on every side; only the rasterizer differs" % ITERS,
           "marks": MARKS, "runs": runs, "wall_s": {k: round(v, 1) for k, v in secs.items()}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "trajectory_parity.json"), "w") as f:
        json.dump(out, f, indent=1)
    p, r, r2 = runs["product"], runs["reference"], runs["reference2"]
    print(json.dumps(out["wall_s"]), {m: (p[m]["psnr"], r[m]["psnr"], r2[m]["psnr"], p[m].get("gaussians"), r[m].get("gaussians"), r2[m].get("gaussians")) for m in MARKS})
    # identical start; before the first densification the three trajectories differ by accumulation-order noise only
    assert abs(p[1]["psnr"] - r[1]["psnr"]) < 0.02 and abs(p[100]["psnr"] - r[100]["psnr"]) < 0.1, (p[1], r[1], p[100], r[100])
    # the whole curve.  The reference's atomics make two runs of the SAME kernels diverge once densification decisions flip: over four
    # reference runs on two GPU boxes the final PSNR ranged 44.79 ... 45.14 dB and the final count 6811 ... 6878 Gaussians
    # (profiles/r04_trajectory_parity.json); the product -- deterministic: 45.355 dB, 6895 Gaussians on both boxes -- is held to the
    # mean of this run's two reference curves within 0.6 dB + their spread, and 3 % + their spread in the number of Gaussians
    # (over five reference-run pairs on four boxes by now: final PSNR 44.79 ... 45.14 dB).
    for m in MARKS[1:]:
        spread_psnr = abs(r[m]["psnr"] - r2[m]["psnr"])
        mean_psnr = 0.5 * (r[m]["psnr"] + r2[m]["psnr"])
        mean_n = 0.5 * (r[m]["gaussians"] + r2[m]["gaussians"])
        spread_n = abs(r[m]["gaussians"] - r2[m]["gaussians"]) / mean_n
        assert abs(p[m]["psnr"] - mean_psnr) <= (0.1 if m <= 250 else 0.6) + spread_psnr, (m, p[m], r[m], r2[m])
        # (m <= 250: one pruning pass behind the start -- 1 %: on a fourth box both reference runs kept 325 Gaussians where every
        # other run, reference or product, kept 326; the count at 100 is the initial 6000 on every side)
        assert abs(p[m]["gaussians"] - mean_n) / mean_n <= (0.01 if m <= 250 else 0.03) + spread_n, (m, p[m], r[m], r2[m])
    assert p[100]["gaussians"] == r[100]["gaussians"] == r2[100]["gaussians"]
    assert p[ITERS]["psnr"] > p[1]["psnr"] + 6.0 and r[ITERS]["psnr"] > r[1]["psnr"] + 6.0
