"""Structural counts of the two blend kernels on a full-size workload, WITHOUT a GPU: the instrumented build (-DGOF_STATS) of the
kernels' source run on the host (tests/hipemu).  The counters are exact properties of the algorithm on the scene (they equal what
the GPU's instrumented build counts: S1M 121 988 948 contributing pairs, 2 647 426 staged entries) and are what the cycle model of
DESIGN.md section 5 is fed with: forward -- candidates / popped / evaluated / contributing pairs per pixel, phase-2 trips per lane,
and what a DECOUPLED dense evaluation would have to evaluate per chunk size; backward -- lanes per visited entry, how often two
consecutive visited entries have disjoint lanes (entry merging), trips of a walk per 16-lane row.
    python tests/devtools/dev_hipemu_structure.py [s1m|s1m_clustered|small] -> gpurun_out/r03_structure_counts.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hipemu"), os.path.join(ROOT, "gaussian-opacity-fields_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import emu_binding as E  # noqa: E402
import synthetic_scenes as S  # noqa: E402

SCENES = {"s1m": lambda: S.scene_frustum(1_000_000, seed=0), "s1m_clustered": lambda: S.scene_clustered(1_000_000, seed=0),
          "small": lambda: S.scene_frustum(150_000, W=640, H=400, focal=480.0, seed=0)}


def run(which, chunks=(256, 64, 32)):
    sc = SCENES[which]()
    npix = sc["W"] * sc["H"]
    out = (C.c_ulonglong * 12)()
    rep = {"scene": which, "P": int(sc["means3D"].shape[0]), "W": sc["W"], "H": sc["H"], "forward": {}}
    for chunk in chunks:
        lib = E.load(extra_flags=("-DGOF_STATS", "-DGOF_FW_CHUNK=%d" % chunk), tag="stats_c%d" % chunk)
        lib.gof_debug_fw_stats(out, 1)
        e = E.EmuScene(sc, lib=lib); pc, _ = e.forward()
        lib.gof_debug_fw_stats(out, 1)
        s = list(out)
        rep["R"] = e.R
        rep["forward"]["chunk_%d" % chunk] = {
            "entries_scanned_per_pixel": s[0] * 64 / npix, "candidates_per_pixel_after_the_per_chunk_saturation_mask": s[1] / npix,
            "popped_per_pixel": s[5] / npix, "passed_the_exact_test_per_pixel": s[3] / npix, "contributing_per_pixel": s[4] / npix,
            "phase2_wave_trips_per_lane": s[2] * 64 / npix, "phase2_lane_utilisation": s[5] / (s[2] * 64),
            "dense_evaluations_per_lane_if_decoupled": s[1] / npix}
        if chunk == chunks[0]:
            lib.gof_debug_bw_stats(out, 1)
            e.backward(np.ones_like(pc))
            lib.gof_debug_bw_stats(out, 1)
            b = list(out)
            rep["backward"] = {"staged_entries": b[4], "visited_wave_entries": b[0], "contributing_lane_pairs": b[2], "lanes_per_visit": b[2] / b[0],
                               "consecutive_visits_with_disjoint_lanes": b[5], "visits_saved_by_merging_fraction": b[5] / b[0],
                               "rows_with_a_contributor_per_visit": b[6] / b[0], "visits_with_at_most_32_lanes_fraction": b[7] / b[0],
                               "trips_if_every_row_walked_its_own_union_fraction": b[1] / b[3],
                               "batches": b[9], "visits_per_batch_mean_over_waves": b[0] / (4.0 * b[9]), "visits_per_batch_of_the_slowest_wave": b[8] / b[9],
                               "slowest_wave_over_mean_per_batch": b[8] * 4.0 / b[0],
                               "slowest_wave_over_mean_per_tile_if_the_waves_ran_without_batch_barriers": b[10] * 4.0 / b[0]}
    return rep


if __name__ == "__main__":
    reps = [run(w) for w in (sys.argv[1:] or ["s1m"])]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_structure_counts.json"), "w") as f:
        json.dump({"tool": "tests/devtools/dev_hipemu_structure.py", "scenes": reps}, f, indent=1)
    print(json.dumps(reps, indent=1))
