"""profiles/rNN_parity_report.md from the JSON tests/devtools/dev_parity_report.py writes.
    python tests/devtools/dev_parity_report_md.py gpurun_out/r05_parity_report.json > profiles/r05_parity_report.md"""
import json, sys
r = json.load(open(sys.argv[1]))
S = r["scenes"]
f = lambda x: "%.1e" % x
print("# Measured parity, round 5 (tests/devtools/dev_parity_report.py on 1x MI355X, HEAD's kernels; full data: r05_parity_report.json)\n")
print("Product = libgof_hip.so through the C ABI in its SHIPPED configuration (forward blend: default mode `pair_nodiv_cc`; backward with the LDS wave "
      "reduction and the pools) and, where stated, in the forward's verification mode; oracle = oracle/gof_oracle.cpp (double accumulation in list "
      "order); reference = the reference's own CUDA source compiled for this GPU (oracle/_ref, no-contraction build, fp32 atomics, worst of 3 runs).\n")
print("`posed` = random SE(3) camera (synthetic_scenes.pose_scene); d/s = median over the visible Gaussians of ||mean_view|| / smallest scale (the "
      "conditioning of min_value is its square, SURVEY section 7); histogram = visible Gaussians with d/s in [0,10) [10,30) [30,100) [100,300) [300,1000) [1000,3000) [3000,inf).\n")
print("## Forward, shipped (default) mode vs oracle\n")
print("Integer state = radii, every K1 float, sort keys, sorted list, ranges, n_contrib (bit for bit); `ch 0-2,6,7` = colour, depth, alpha. Channel 8 (distortion): "
      "max |a - ref|, the same over the channel's maximum, 99.9th percentile of the element-wise relative error (|ref| > 1e-3 max).\n")
print("| scene | P | R | ks | posed | d/s median | d/s histogram | integer state | ch 0-2,6,7 bit-equal | normals max abs | ch 8 max | ch 8 max abs | / max | p99.9 elem | decisions = verification mode | verification mode ch 8 bit-equal |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for n, s in S.items():
    fw = s["forward"]; d = s["forward_default_mode_vs_oracle"]; x = s["forward_verification_mode_vs_oracle"]; m = s["default_vs_verification_mode"]
    ints = all(v for k, v in fw.items() if k.endswith("_equal") and k != "final_T_bit_equal")      # (final_T's planes 1-3 are dist1 / dist2 / distortion: with channel 8)
    print("| %s | %d | %d | %.1f | %s | %.0f | %s | %s | %.6f | %s | %s | %s | %s | %s | %s | %.4f |" % (
        n, s["P"], s["R"], s["kernel_size"], "yes" if s["posed"] else "no", s["distance_over_scale"]["median"], " ".join(str(c) for c in s["distance_over_scale"]["visible_count"]),
        "bit-exact" if ints else "DIFFERS", min(d[str(c)]["bit_equal_fraction"] for c in (0, 1, 2, 6, 7)), f(fw["normals_ch_3_4_5_max_abs"]),
        f(d["8"]["ref_max"]), f(d["8"]["max_abs"]), f(d["8"]["max_abs_over_ref_max"]), f(d["8"]["p999_elem"]), "yes" if m["decisions_equal"] else "NO", x["8"]["bit_equal_fraction"]))
print("\n## Blend backward (K8) vs oracle: max-norm error / relative L2 / 99.9th percentile element-wise relative error (|ref| > 1e-3 max)\n")
print("| scene | dL_dmeans2D | dL_dcolors | dL_dopacity | dL_dview2gaussian |\n|---|---|---|---|---|")
for n, s in S.items():
    b = s["blend_backward_vs_oracle"]
    print("| %s | %s |" % (n, " | ".join("%s / %s / %s" % (f(b[k]["max_norm"]), f(b[k]["rel_l2"]), f(b[k]["p999_elem"])) for k in ("means2D", "colors", "opacity", "view2gaussian"))))
print("\n## Per-Gaussian backward (K9) on identical inputs: max-norm error vs oracle\n")
print("| scene | dL_dmeans3D | dL_dsh | dL_dscales | dL_drotations |\n|---|---|---|---|---|")
for n, s in S.items():
    b = s["per_gaussian_backward_on_identical_inputs"]
    print("| %s | %s |" % (n, " | ".join(f(b[k]["max_norm"]) for k in ("means3D", "sh", "scales", "rotations"))))
print("\n## End-to-end parameter gradients vs oracle, relative L2: product / reference (worst of 3 runs) / reference run-to-run; `wc` = well-conditioned subset (d/s <= 30)\n")
print("| scene | dL_dmeans3D | dL_dscales | dL_drotations | dL_dsh | dL_dopacity | wc: dL_dscales product / reference |\n|---|---|---|---|---|---|---|")
for n, s in S.items():
    e = s["end_to_end_parameter_gradients"]
    def cell(k):
        v = e[k]
        t = f(v["product_vs_oracle"]["rel_l2"])
        if "reference_vs_oracle_worst_of_3_runs" in v:
            t += " / %s / %s" % (f(v["reference_vs_oracle_worst_of_3_runs"]["rel_l2"]), f(v["reference_run_to_run"]["rel_l2"]))
        return t
    wc = "-"
    v = e["scales"]
    if "product_vs_oracle_well_conditioned" in v:
        wc = f(v["product_vs_oracle_well_conditioned"]["rel_l2"]) + (" / " + f(v["reference_vs_oracle_well_conditioned"]["rel_l2"]) if "reference_vs_oracle_well_conditioned" in v else "")
    print("| %s | %s | %s |" % (n, " | ".join(cell(k) for k in ("means3D", "scales", "rotations", "sh", "opacity")), wc))
print("\n## Worst figure per tensor over all scenes\n\n| figure | value | scene |\n|---|---|---|")
for k, v in r.get("worst", {}).items():
    print("| %s | %s | %s |" % (k, f(v["value"]), v["scene"]))
