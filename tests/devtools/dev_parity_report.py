"""Measured parity, written down (round-2 review: "commit the parity numbers"; SURVEY.md section 7 hard part (3): "publish, next
to every parity number, the distance/scale histogram of the test scene and a well-conditioned-subset figure").

For every scene of the list below -- the scene table of tests/test_parity_gpu.py incl. the posed scenes, some fuzz seeds, S1M and
posed S1M -- run the product (libgof_hip.so through the C ABI), the oracle and (for the end-to-end parameter gradients) the
reference's own CUDA source compiled for this GPU (oracle/_ref, no-contraction build, three runs) on the same seeded inputs and
record, per tensor:

  max_norm   max|a - ref| / max|ref|                      (what the tests assert)
  rel_l2     ||a - ref||_2 / ||ref||_2
  p999_elem  99.9th percentile of |a - ref| / |ref| over the elements with |ref| > 1e-3 max|ref|   (element-wise relative error)

plus, for the scene, the histogram of distance/scale (||mean_view|| / smallest scale: the conditioning of min_value = CC -
BB^2/(4 AA) is its square) over the visible Gaussians, and the same error figures restricted to the
well-conditioned subset (ratio <= 30) of the Gaussians.  Forward: fraction of bit-identical elements per channel group.

Writes gpurun_out/r05_parity_report.json (copied to profiles/ by hand).  GPU box only (test infrastructure: uses the oracle).
    python tests/devtools/dev_parity_report.py [--quick]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gpu_common import bits, fetch, product_forward_raw, to_dev, ROOT  # noqa: E402
import oracle_binding as ob  # noqa: E402
import reference_binding as rb  # noqa: E402
import synthetic_scenes as S  # noqa: E402
import test_parity_gpu as TP  # noqa: E402

RATIO_BINS = [0, 10, 30, 100, 300, 1000, 3000, 1e30]


def err(a, ref):
    a = np.asarray(a, np.float64).ravel(); ref = np.asarray(ref, np.float64).ravel()
    m = np.abs(ref).max() if ref.size else 0.0
    if m == 0.0:
        return {"max_norm": float(np.abs(a).max()) if a.size else 0.0, "rel_l2": 0.0, "p999_elem": 0.0, "n_elem": 0, "ref_max": 0.0}
    d = np.abs(a - ref)
    big = np.abs(ref) > 1e-3 * m
    return {"max_norm": float(d.max() / m), "rel_l2": float(np.linalg.norm(a - ref) / np.linalg.norm(ref)),
            "p999_elem": float(np.percentile(d[big] / np.abs(ref[big]), 99.9)) if big.any() else 0.0,
            "p50_elem": float(np.percentile(d[big] / np.abs(ref[big]), 50)) if big.any() else 0.0,
            "n_elem": int(big.sum()), "ref_max": float(m)}


def ratios(sc):
    """distance / smallest scale per Gaussian, in view space"""
    V = sc["viewmatrix"].astype(np.float64)
    mv = sc["means3D"].astype(np.float64) @ V[:3, :3] + V[3, :3]
    return np.linalg.norm(mv, axis=1) / np.maximum(sc["scales"].astype(np.float64).min(1), 1e-30)


def report_scene(name, sc, with_reference):
    t0 = time.time()
    o, oc, orad, res = TP._forward_pair(sc)
    P = len(orad); vis = orad > 0
    out = {"P": P, "W": sc["W"], "H": sc["H"], "R": int(res["R"]), "visible": int(vis.sum()), "kernel_size": sc["kernel_size"],
           "scale_modifier": sc["scale_modifier"], "sh_degree": sc["sh_degree"],
           "posed": bool(np.abs(sc["viewmatrix"][:3, :3] - np.eye(3)).max() > 1e-6 or np.abs(sc["campos"]).max() > 0)}
    r = ratios(sc)
    out["distance_over_scale"] = {"bins": RATIO_BINS[:-1], "visible_count": np.histogram(r[vis], RATIO_BINS)[0].tolist(),
                                  "median": float(np.median(r[vis])) if vis.any() else 0.0, "p99": float(np.percentile(r[vis], 99)) if vis.any() else 0.0}
    well = vis & (r <= 30.0)
    out["well_conditioned_visible"] = int(well.sum())
    # forward
    pc = res["color"].cpu().numpy()
    fw = {"radii_equal": bool(np.array_equal(res["radii"].cpu().numpy(), orad)), "R_equal": bool(res["R"] == o.num_rendered())}
    for arr in TP.K1_ARRAYS:
        a = fetch(res, arr).reshape(P, -1)[vis]; b = o.fetch(arr).reshape(P, -1)[vis]
        if a.dtype != b.dtype:
            a = a.view(b.dtype) if a.itemsize == b.itemsize else a.astype(b.dtype)
        fw["K1_" + arr + "_bit_equal"] = bool(np.array_equal(bits(a), bits(b)))
    for arr in TP.INT_ARRAYS:
        fw[arr + "_bit_equal"] = bool(TP._same(fetch(res, arr), o.fetch(arr)))
    fw["final_T_bit_equal"] = bool(np.array_equal(bits(fetch(res, "final_T")), bits(o.fetch("final_T"))))
    ex = TP.EXACT_CH
    fw["image_ch_0_1_2_6_7_8_bit_equal_fraction"] = float((bits(pc[ex]) == bits(oc[ex])).mean())
    fw["image_ch_0_1_2_6_7_8_max_abs"] = float(np.abs(pc[ex] - oc[ex]).max())
    fw["normals_ch_3_4_5_max_abs"] = float(np.abs(pc[3:6] - oc[3:6]).max())
    out["forward"] = fw
    # round 5: the SHIPPED (default) forward mode and the verification mode, channel by channel against the oracle: fraction of
    # bit-identical pixels, max |a - ref|, the same relative to the channel's maximum (north_star's "1e-4 relative"), and the 99.9th
    # percentile of the element-wise relative error over the pixels with |ref| > 1e-3 max|ref|
    def per_channel(img):
        rows = {}
        for ch in range(9):
            a = img[ch].astype(np.float64).ravel(); b = oc[ch].astype(np.float64).ravel()
            m = float(np.abs(b).max())
            d = np.abs(a - b)
            big = np.abs(b) > 1e-3 * m if m > 0 else np.zeros_like(b, bool)
            rows[str(ch)] = {"bit_equal_fraction": float((bits(img[ch]) == bits(oc[ch])).mean()), "ref_max": m, "max_abs": float(d.max()),
                             "max_abs_over_ref_max": float(d.max() / m) if m > 0 else 0.0,
                             "p999_elem": float(np.percentile(d[big] / np.abs(b[big]), 99.9)) if big.any() else 0.0}
        return rows
    out["forward_default_mode_vs_oracle"] = per_channel(pc)
    out["forward_verification_mode_vs_oracle"] = per_channel(res["exact"]["color"].cpu().numpy())
    px = res["exact"]["color"].cpu().numpy()
    d8 = np.abs(pc[8].astype(np.float64) - px[8])
    out["default_vs_verification_mode"] = {"decisions_equal": bool(np.array_equal(fetch(res, "n_contrib"), fetch(res["exact"], "n_contrib")) and
                                                                    np.array_equal(fetch(res, "contrib_hash"), fetch(res["exact"], "contrib_hash"))),
                                           "ch_0_7_bit_equal_fraction": float((bits(pc[:8]) == bits(px[:8])).mean()),
                                           "ch8_max_abs": float(d8.max()), "ch8_ref_max": float(np.abs(px[8]).max()),
                                           "ch8_max_abs_over_ref_max": float(d8.max() / max(float(np.abs(px[8]).max()), 1e-300))}
    # backward
    dL = np.random.default_rng(17).normal(size=oc.shape).astype(np.float32)
    go = o.backward(dL)
    gp = TP._product_backward(res, dL)
    bw = {}
    for k in ("means2D", "colors", "opacity", "view2gaussian"):
        bw[k] = err(gp[k].reshape(go[k].shape), go[k])
        if well.any():
            bw[k]["well_conditioned"] = err(gp[k].reshape(P, -1)[well], go[k].reshape(P, -1)[well])
    out["blend_backward_vs_oracle"] = bw
    iso = o.preprocess_backward(gp["view2gaussian"], gp["colors"])
    out["per_gaussian_backward_on_identical_inputs"] = {k: err(gp[k].reshape(iso[k].shape), iso[k]) for k in ("means3D", "sh", "scales", "rotations")}
    e2e = {}
    runs = None
    if with_reference:
        ref = rb.Reference(to_dev(sc), "_nofma")
        ref.forward()
        runs = [ref.backward(dL) for _ in range(3)]
    for k in ("means3D", "scales", "rotations", "sh", "opacity"):
        e = {"product_vs_oracle": err(gp[k].reshape(go[k].shape), go[k])}
        if well.any():
            e["product_vs_oracle_well_conditioned"] = err(gp[k].reshape(P, -1)[well], go[k].reshape(P, -1)[well])
        if runs is not None:
            per = [err(r_[k].reshape(go[k].shape), go[k]) for r_ in runs]
            e["reference_vs_oracle_worst_of_3_runs"] = {m: max(p[m] for p in per) for m in ("max_norm", "rel_l2", "p999_elem")}
            spread = [err(runs[i][k], runs[0][k]) for i in (1, 2)]
            e["reference_run_to_run"] = {m: max(p[m] for p in spread) for m in ("max_norm", "rel_l2", "p999_elem")}
            if well.any():
                perw = [err(r_[k].reshape(P, -1)[well], go[k].reshape(P, -1)[well]) for r_ in runs]
                e["reference_vs_oracle_well_conditioned"] = {m: max(p[m] for p in perw) for m in ("max_norm", "rel_l2", "p999_elem")}
        e2e[k] = e
    out["end_to_end_parameter_gradients"] = e2e
    out["seconds"] = round(time.time() - t0, 1)
    print(name, "R=%d" % out["R"], "fwd exact frac %.6f" % fw["image_ch_0_1_2_6_7_8_bit_equal_fraction"],
          {k: "%.1e" % v["max_norm"] for k, v in bw.items()}, {k: "%.1e" % v["product_vs_oracle"]["rel_l2"] for k, v in e2e.items()}, flush=True)
    return out


def main():
    quick = "--quick" in sys.argv
    table = {k: TP.SCENES[k] for k in TP.SCENES}
    for seed in (101, 102, 103, 105, 106, 107):
        table["fuzz_%d" % seed] = (lambda s=seed: TP._fuzz_scene(s))
    if not quick:
        table["s1m"] = lambda: S.scene_frustum(1_000_000, seed=0)
        table["s1m_posed"] = lambda: S.scene_frustum(1_000_000, seed=0, pose_seed=0)
        table["s1m_clustered"] = lambda: S.scene_clustered(1_000_000, seed=0)
        table["s1m_ks01"] = lambda: S.scene_frustum(1_000_000, seed=0, kernel_size=0.1)
        table["s1m_posed_ks01"] = lambda: S.scene_frustum(1_000_000, seed=0, pose_seed=0, kernel_size=0.1)
    rep = {"tool": "tests/devtools/dev_parity_report.py", "gpu": torch.cuda.get_device_name(0),
           "definitions": {"max_norm": "max|a-ref| / max|ref|", "rel_l2": "||a-ref|| / ||ref||",
                           "p999_elem": "99.9th percentile of |a-ref|/|ref| over elements with |ref| > 1e-3 max|ref|",
                           "distance_over_scale": "||mean in view space|| / smallest scale, visible Gaussians; well-conditioned = ratio <= 30",
                           "oracle": "oracle/gof_oracle.cpp (double accumulation in list order); reference = oracle/_ref no-contraction build of the reference's own CUDA source, fp32 atomics"},
           "scenes": {}}
    for name, mk in table.items():
        sc = mk()
        P = sc["means3D"].shape[0]
        rep["scenes"][name] = report_scene(name, sc, with_reference=rb.available("_nofma") and P >= 100)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_parity_report.json"), "w") as f:
        json.dump(rep, f, indent=1)
    # summary: worst figure per tensor over all scenes
    worst = {}
    for sn, s in rep["scenes"].items():
        for grp in ("blend_backward_vs_oracle", "per_gaussian_backward_on_identical_inputs"):
            for k, v in s[grp].items():
                for m in ("max_norm", "rel_l2", "p999_elem"):
                    key = "%s.%s.%s" % (grp, k, m)
                    if v[m] > worst.get(key, (0, ""))[0]:
                        worst[key] = (v[m], sn)
    for sn, s in rep["scenes"].items():          # round 5: the shipped forward mode's distortion channel, relative to the channel's maximum
        v = s["forward_default_mode_vs_oracle"]["8"]
        for m in ("max_abs", "max_abs_over_ref_max", "p999_elem"):
            key = "forward_default_mode_vs_oracle.ch8.%s" % m
            if v[m] > worst.get(key, (0, ""))[0]:
                worst[key] = (v[m], sn)
    rep["worst"] = {k: {"value": v[0], "scene": v[1]} for k, v in worst.items()}
    with open(os.path.join(ROOT, "gpurun_out", "r05_parity_report.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({"worst": worst}, indent=1))


if __name__ == "__main__":
    main()
