"""BASELINE's headline configuration (S1M: 1M Gaussians @ 1600x1063, SH degree 3) through the kernels' SOURCE run on the host
(tests/hipemu) against the oracle -- the full-size parity check of tests/test_parity_gpu.py::test_full_size_s1m_* without a GPU.
Minutes on 8 cores; writes gpurun_out/r04_hipemu_full_size.json (copied to profiles/ by hand).
    python tests/devtools/dev_hipemu_full_size.py [s1m] [s1m_posed] [s1m_clustered]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hipemu"), os.path.join(ROOT, "gaussian-opacity-fields_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import emu_binding as E  # noqa: E402
import oracle_binding as ob  # noqa: E402
import synthetic_scenes as S  # noqa: E402
import test_parity_gpu as TP  # noqa: E402
from gpu_common import bits  # noqa: E402

SCENES = {"s1m": lambda: S.scene_frustum(1_000_000, seed=0), "s1m_posed": lambda: S.scene_frustum(1_000_000, seed=0, pose_seed=0),
          "s1m_clustered": lambda: S.scene_clustered(1_000_000, seed=0)}

if __name__ == "__main__":
    rep = {}
    for name in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["s1m"]):
        sc = SCENES[name]()
        t = time.time(); o = ob.OracleScene(sc); oc, orad = o.forward(); t_of = time.time() - t
        e = E.EmuScene(sc)                                    # the default mode (what ships): decisions, integer arrays, the backward
        t = time.time(); pf, prad = e.forward(); t_ef = time.time() - t
        ex = E.EmuScene(sc, exact=True)                       # the verification mode: the image's bits
        pc, _ = ex.forward()
        P = len(orad); vis = orad > 0
        r = {"P": P, "W": sc["W"], "H": sc["H"], "R": e.R, "R_equal": e.R == o.num_rendered(), "radii_equal": bool(np.array_equal(prad, orad)),
             "seconds": {"oracle_forward": round(t_of, 1), "emulated_forward": round(t_ef, 1)}}
        for arr in TP.K1_ARRAYS:
            a = e.fetch(arr).reshape(P, -1)[vis]; b = o.fetch(arr).reshape(P, -1)[vis]
            r["K1_%s_bit_equal" % arr] = bool(TP._same(a, b))
        for arr in TP.INT_ARRAYS:
            r["%s_bit_equal" % arr] = bool(TP._same(e.fetch(arr), o.fetch(arr)))
        r["final_T_bit_equal_verification_mode"] = bool(np.array_equal(bits(ex.fetch("final_T")), bits(o.fetch("final_T"))))
        r["image_ch_0_1_2_6_7_8_bit_equal_verification_mode"] = bool(np.array_equal(bits(pc[TP.EXACT_CH]), bits(oc[TP.EXACT_CH])))
        HW = sc["W"] * sc["H"]
        r["default_mode"] = {"T_bit_equal": bool(np.array_equal(bits(e.fetch("final_T")[:HW]), bits(o.fetch("final_T")[:HW]))),
                             "pixels_with_different_bits_ch_0_1_2_6_7": [int((bits(pf[c]) != bits(oc[c])).sum()) for c in (0, 1, 2, 6, 7)],
                             "distortion_channel_max_abs": float(np.abs(pf[8] - oc[8]).max())}
        r["normals_max_abs"] = float(np.abs(pc[3:6] - oc[3:6]).max())
        dL = np.random.default_rng(17).normal(size=oc.shape).astype(np.float32)
        t = time.time(); go = o.backward(dL); t_ob = time.time() - t
        t = time.time(); gp = e.backward(dL); t_eb = time.time() - t
        r["seconds"].update(oracle_backward=round(t_ob, 1), emulated_backward=round(t_eb, 1))
        r["blend_backward_max_norm_error"] = {k: float(np.abs(gp[k].reshape(go[k].shape) - go[k]).max() / (np.abs(go[k]).max() + 1e-30)) for k in ("means2D", "colors", "opacity", "view2gaussian")}
        iso = o.preprocess_backward(gp["view2gaussian"], gp["colors"])
        r["K9_on_identical_inputs_max_norm_error"] = {k: float(np.abs(gp[k].reshape(iso[k].shape) - iso[k]).max() / (np.abs(iso[k]).max() + 1e-30)) for k in ("means3D", "sh", "scales", "rotations")}
        if "--integrate" in sys.argv:      # the opacity-field query: 1M of the scene's 9M tetra points
            pts = S.tetra_points(sc)
            pts = np.ascontiguousarray(pts[np.random.default_rng(3).choice(len(pts), 1_000_000, replace=False)], dtype=np.float32)
            t = time.time(); ic, ia, icol, irad = ob.OracleScene(sc).integrate(pts); t_oi = time.time() - t
            t = time.time(); c2, a2, col2, rad2 = E.EmuScene(sc).integrate(pts); t_ei = time.time() - t
            r["integrate_1M_points"] = {"image_bit_equal": bool(np.array_equal(bits(c2), bits(ic))), "alpha_integrated_bit_equal": bool(np.array_equal(bits(a2), bits(ia))),
                                        "color_integrated_bit_equal": bool(np.array_equal(bits(col2), bits(icol))), "radii_equal": bool(np.array_equal(rad2, irad)),
                                        "seconds": {"oracle": round(t_oi, 1), "emulated": round(t_ei, 1)}}
        r["guards_intact"] = E.guards_intact() == []
        rep[name] = r
        print(name, json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    prev = os.path.join(ROOT, "profiles", "r04_hipemu_full_size.json")
    if os.path.exists(prev):                       # scenes not re-run keep their committed entries
        old = json.load(open(prev)).get("scenes", {})
        rep = {**old, **rep}
    with open(os.path.join(ROOT, "gpurun_out", "r04_hipemu_full_size.json"), "w") as f:
        json.dump({"tool": "tests/devtools/dev_hipemu_full_size.py", "what": "csrc/*.hip compiled for the host (tests/hipemu) vs the oracle at BASELINE's full size, no GPU", "scenes": rep}, f, indent=1)
