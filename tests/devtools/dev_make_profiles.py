"""Developer script: turn the rocprofv3 outputs merged under gpurun_out/ into the committed summaries in profiles/.
usage: python tests/devtools/dev_make_profiles.py <round e.g. r02> <tag e.g. v2> <kernel-trace stats dir> <pmc root dir> [bench json]
The pmc root holds one sub-directory per counter pass (tests/devtools/dev_r2_profiles.sh): fetch (FETCH_SIZE), write (WRITE_SIZE),
sq1 (SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_*), sq2 (SQ_WAIT_* SQ_INSTS_* SQ_THREAD_CYCLES_VALU), sq3 (LDS + GRBM_GUI_ACTIVE),
sq4 (SQ_INSTS_VALU_* by type).  Writes profiles/<round>_bench_s1m_kernel_stats_<tag>.{csv,md}, profiles/<round>_pmc_traffic_s1m.json
(keyed by kernel; "_kernel_sha16" = hash of the blend kernel sources the passes ran on, bench.py refuses a stale file) and copies
the bench line."""
import csv, glob, json, os, re, sys, shutil
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rnd, tag, prof, pmc_root = sys.argv[1:5]
bench = sys.argv[5] if len(sys.argv) > 5 else None


def short(name):
    m = re.match(r"(?:void )?(?:gof::)?([A-Za-z0-9_]+)", name)
    n = m.group(1) if m else name
    return {"__amd_rocclr_fillBufferAligned": "hipMemsetAsync (fillBufferAligned)", "__amd_rocclr_copyBuffer": "hipMemcpyAsync (copyBuffer)"}.get(n, n)


def find(d, suffix):
    fs = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None


stats_file = find(prof, "kernel_stats.csv")
stats = list(csv.DictReader(open(stats_file)))
shutil.copy(stats_file, os.path.join(ROOT, "profiles", "%s_bench_s1m_kernel_stats_%s.csv" % (rnd, tag)))
agg = defaultdict(lambda: [0, 0.0])
for r in stats:
    k = short(r["Name"])
    if k.startswith("at") or "elementwise" in r["Name"]:
        k = "torch elementwise (zeros/fill)"
    agg[k][0] += int(r["Calls"]); agg[k][1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())


def pmc(sub):
    f = find(os.path.join(pmc_root, sub), "counter_collection.csv")
    out = defaultdict(lambda: defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


passes = {s: pmc(s) for s in ("fetch", "write", "sq1", "sq2", "sq3", "sq4", "tcc")}
kernels = sorted(set().union(*[set(p) for p in passes.values()]))
import bench as bench_mod
data = {"_kernel_sha16": bench_mod.kernel_sha16(), "_sha16_by_kernel": {k: bench_mod.kernel_sha16(k) for k in kernels if k in bench_mod.KERNEL_SOURCES},
        "_note": "per-launch averages; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md); "
                 "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves / SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
for k in kernels:
    if k.startswith("hipMem") or k.startswith("torch") or k.startswith("at"):
        continue
    e = {}
    f, w = passes["fetch"].get(k, {}).get("FETCH_SIZE"), passes["write"].get(k, {}).get("WRITE_SIZE")
    if f is not None or w is not None:
        e.update(fetch_KiB_raw=f or 0.0, write_KiB=w or 0.0, hbm_bytes_corrected=2 * (f or 0.0) * 1024 + (w or 0.0) * 1024)
    c = {}
    for s in ("sq1", "sq2", "sq3", "sq4", "tcc"):
        c.update(passes[s].get(k, {}))
    if c:
        e["counters"] = c
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui and c.get("SQ_ACTIVE_INST_VALU"):
            cycles = gui / 8.0
            e["gpu_cycles"] = cycles
            # rocprof's VALUBusy: per-WAVE "a VALU instruction of mine is in the pipe" time (quad-cycles -> cycles) summed over the waves,
            # over the SIMD-cycles of the launch.  Two waves of a SIMD overlap in the pipe (the next wave issues while a multi-pass
            # instruction -- fp64, transcendental, cross-lane -- of another still executes), so the ratio is an ACTIVITY figure that can
            # exceed 1 (integrate_pixels 1.10, integrate_points 1.43 in round 2: long fp32 division / sqrt sequences); it was labelled
            # "pipe busy" before, which a value above 1 cannot be.  The bounded companion is the issue rate below.
            e["valu_active_over_simd_cycles"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles)
            e["valu_pipe_busy_frac"] = e["valu_active_over_simd_cycles"]                          # (old key, kept for bench.py's reader)
            e["valu_issue_frac"] = min(1.0, e["valu_active_over_simd_cycles"])
            if c.get("SQ_INSTS_VALU"):
                e["valu_insts_per_simd_cycle"] = c["SQ_INSTS_VALU"] / (1024.0 * cycles)           # wave64 instructions issued per SIMD and cycle
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            e["cycles_per_valu_inst"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
        if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            e["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        if c.get("SQ_WAVE_CYCLES"):
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c.get(n) is not None:
                    e[n.lower() + "_frac_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    if e:
        data[k] = e
json.dump(data, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic_s1m.json" % rnd), "w"), indent=1)
md = os.path.join(ROOT, "profiles", "%s_bench_s1m_kernel_stats_%s.md" % (rnd, tag))
with open(md, "w") as o:
    o.write("# rocprofv3 summaries, round %s, kernels of commit-state '%s' (blend kernel sources sha16 %s)\n\n" % (rnd[1:], tag, data["_kernel_sha16"]))
    o.write("`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-integrate --no-full-loop --no-clustered --no-views --no-reference --no-kernel-size-leg --no-large-p`\n"
            "(S1M: 1M Gaussians, 1600x1063, R = 8 837 593)\n\n")
    o.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write("| %s | %d | %.3f | %.1f | %.2f |\n" % (k, c, t / 1e6, t / c / 1e3, 100 * t / tot))
    o.write("\n## HBM traffic per launch (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes over tests/devtools/dev_pmc.py)\n\n"
            "FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at 1/2 (MI355X guide), so the read\n"
            "column is given raw and doubled; the 64 MB record table and the tile lists stay resident in the 256 MiB Infinity Cache.\n\n")
    o.write("| kernel | FETCH_SIZE KiB (raw) | read MB (x2 corrected) | WRITE_SIZE KiB | write MB |\n|---|---|---|---|---|\n")
    for k, v in sorted(((k, v) for k, v in data.items() if isinstance(v, dict) and "hbm_bytes_corrected" in v), key=lambda kv: -kv[1]["hbm_bytes_corrected"]):
        o.write("| %s | %.0f | %.1f | %.0f | %.1f |\n" % (k, v["fetch_KiB_raw"], 2 * v["fetch_KiB_raw"] * 1024 / 1e6, v["write_KiB"], v["write_KiB"] * 1024 / 1e6))
    o.write("\n## Vector-ALU counters of the kernels (separate SQ passes)\n\n"
            "VALU active = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x kernel cycles): per-wave in-pipe time summed over the waves -- waves of a SIMD overlap\n"
            "in the pipe, so this is an activity figure that CAN exceed 1 (it is rocprof's VALUBusy); VALU inst / SIMD-cycle = SQ_INSTS_VALU / (1024 x kernel cycles) is the bounded\n"
            "issue rate; cycles per VALU instruction = active time / SQ_INSTS_VALU;\n"
            "lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); wave-cycle split: parked at s_waitcnt / barrier (WAIT_ANY), issue stall\n"
            "(WAIT_INST_ANY), issuing (ACTIVE_INST_ANY).\n\n"
            "| kernel | VALU insts | VALU active / SIMD-cycles | VALU inst / SIMD-cycle | cycles / VALU inst | lane utilisation | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | LDS bank conflict cycles / LDS active | L2 hit rate (TCC_HIT / (HIT + MISS)) |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for k, v in sorted(((k, v) for k, v in data.items() if isinstance(v, dict) and "valu_pipe_busy_frac" in v), key=lambda kv: -kv[1]["counters"].get("SQ_INSTS_VALU", 0)):
        c = v["counters"]
        o.write("| %s | %.3e | %.2f | %.3f | %.2f | %s | %s | %s | %s | %s | %s |\n" % (
            k, c.get("SQ_INSTS_VALU", 0), v["valu_pipe_busy_frac"], v.get("valu_insts_per_simd_cycle", float("nan")), v.get("cycles_per_valu_inst", float("nan")),
            "%.2f" % v["valu_lane_utilisation"] if "valu_lane_utilisation" in v else "-",
            *["%.2f" % v[n] if n in v else "-" for n in ("sq_wait_any_frac_of_wave_cycles", "sq_wait_inst_any_frac_of_wave_cycles", "sq_active_inst_any_frac_of_wave_cycles")],
            "%.2f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else "-",
            "%.3f" % v["l2_hit_rate"] if "l2_hit_rate" in v else "-"))
    o.write("\n## VALU instruction mix of the blend kernels (SQ_INSTS_VALU_* pass)\n\n| kernel | " + " | ".join(
        n.replace("SQ_INSTS_VALU_", "") for n in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                                                 "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")) + " |\n|---|" + "---|" * 8 + "\n")
    for k in ("blend_forward", "blend_backward", "integrate_rays", "integrate_pixels", "integrate_points"):
        c = data.get(k, {}).get("counters", {})
        if c.get("SQ_INSTS_VALU_ADD_F32") is not None:
            o.write("| %s | " % k + " | ".join("%.3e" % c.get(n, 0) for n in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                                                                                "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")) + " |\n")
if bench:
    shutil.copy(bench, os.path.join(ROOT, "profiles", "%s_bench_s1m_%s.json" % (rnd, tag)))
print(open(md).read())
