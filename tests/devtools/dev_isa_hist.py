"""Developer script: instruction-class histogram of the hot loops of the two blend kernels, from the gfx950 assembly the library's
flags produce (no GPU needed).  A loop = the basic blocks LLVM annotates with `in Loop: Header=BBn` (+ the header itself); the hot
loop of a kernel = the innermost loop that contains a marker instruction.  Cycle weights: tests/devtools/microbench (measured on
MI355X, cycles a wave64 instruction occupies its SIMD): profiles/r02_microbench_valu_issue_cycles.txt."""
import os, re, subprocess, sys, collections
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "..", "gaussian-opacity-fields_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         "-fno-slp-vectorize", "--cuda-device-only", "-S"]
COST = [  # (regex on the mnemonic, class, cycles per wave instruction)
    (r"v_rcp_f64|v_rsq_f64|v_sqrt_f64", "fp64 transcendental", 16.4),
    (r"v_.*_f64|v_cvt_f64|v_cvt_f32_f64|v_ldexp_f64|v_div_", "fp64", 5.0),
    (r"v_pk_", "packed fp32", 5.2),
    (r"v_exp_f32|v_rcp_f32|v_rsq_f32|v_log_f32|v_sqrt_f32", "fp32 transcendental", 8.4),
    (r"v_permlane", "lane-group swap", 12.0),
    (r".*_dpp$", "DPP", 4.4),
    (r"v_cndmask|v_cmp|v_alignbit|v_bfe|v_lshl_or|v_lshl_add|v_and_or|v_bitop|v_ffb|v_readlane|v_readfirstlane|v_cvt|v_max|v_min|v_med3|v_ldexp|v_rndne|v_mad_u|v_mul_lo|v_mul_hi", "VOP3 / compare / select / convert", 4.4),
    (r"v_fma_f32|v_fmac|v_fmaak|v_fmamk|v_add_f32|v_sub_f32|v_mul_f32|v_subrev_f32", "plain fp32", 3.0),
    (r"v_mov|v_add_u32|v_sub_u32|v_and_b32|v_or_b32|v_xor|v_lshlrev|v_lshrrev|v_not|v_add_co|v_addc|v_subrev|v_ashrrev|v_accvgpr", "integer / move", 3.0),
    (r"v_", "other VALU", 4.4),
    (r"ds_", "LDS", 0.0), (r"global_|buffer_|flat_", "global memory", 0.0), (r"s_waitcnt|s_nop", "wait / nop", 0.0), (r"s_", "scalar", 0.0),
]
def classify(m):
    base = re.sub(r"_e(32|64)$", "", m)
    for rx, cls, cyc in COST:
        if re.match(rx, m) or re.match(rx, base):
            return cls, cyc
    return "other", 0.0
def kernel_body(asm, name):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3gof\d+%sE[^:]*:" % name, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]
def loops(body):
    # block label -> (header it belongs to or itself, line range)
    blocks, cur = [], None
    for i, l in enumerate(body):
        m = re.match(r"^\.LBB\d+_(\d+):\s*;?(.*)$", l)
        if m:
            cur = {"id": int(m.group(1)), "note": m.group(2), "lines": []}
            blocks.append(cur)
        elif cur is not None and re.match(r"^\s+[a-z]", l):
            cur["lines"].append(l.split()[0])
    out = collections.OrderedDict()
    for b in blocks:
        h = re.search(r"Loop Header: Depth=(\d+)", b["note"])
        inl = re.search(r"in Loop: Header=BB\d+_(\d+) Depth=(\d+)", b["note"])
        par = re.search(r"Parent Loop BB\d+_(\d+) Depth=(\d+)", b["note"])
        key = None
        if inl: key = int(inl.group(1))
        elif h or par:
            key = b["id"]
        if key is not None:
            out.setdefault(key, []).extend(b["lines"])
    return out
def report(src, kernel, marker, title, extra=()):
    asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-o", "-", os.path.join(CSRC, src)], capture_output=True, text=True, check=True).stdout
    body = kernel_body(asm, kernel)
    cand = [(len(v), k, v) for k, v in loops(body).items() if any(re.match(marker, m) for m in v)]
    n, k, ins = min(cand)
    hist = collections.OrderedDict()
    for m in ins:
        cls, cyc = classify(m)
        a = hist.setdefault(cls, [0, 0.0]); a[0] += 1; a[1] += cyc
    valu = sum(c for cls, (c, t) in hist.items() if t > 0)
    cycles = sum(t for c, t in hist.values())
    print("### %s\n\n%s, innermost loop containing `%s` (header BB%d): %d instructions, %d of them VALU, ~%.0f SIMD cycles per trip at the measured "
          "per-class costs\n\n| class | instructions | est. cycles |\n|---|---|---|" % (title, kernel, marker.replace('|', ' / '), k, len(ins), valu, cycles))
    for cls, (c, t) in sorted(hist.items(), key=lambda kv: -kv[1][1] * 1000 - kv[1][0]):
        print("| %s | %d | %s |" % (cls, c, ("%.0f" % t) if t else "-"))
    top = collections.Counter(re.sub(r"_e(32|64)$", "", m) for m in ins).most_common(14)
    print("\nmost frequent: " + ", ".join("%s x%d" % kv for kv in top) + "\n")
if __name__ == "__main__":
    print("# Instruction mix of the hot loops (gfx950 assembly of the shipped sources; tests/devtools/dev_isa_hist.py)\n")
    extra = sys.argv[1:]          # e.g. -DGOF_FW_EXACT, -DGOF_FW_HWEXP: the variant builds
    if extra:
        print("build flags: `%s`\n" % " ".join(extra))
    # (marker v_rsq_f32: the unit normal, present in the phase-2 / gradient loop of every variant)
    report("blend_forward.hip", "blend_forward", r"v_rsq_f32", "blend_forward, phase 2: one candidate (pixel, entry) pair per lane and trip", extra)
    report("blend_forward.hip", "blend_forward", r"v_alignbit", "blend_forward, phase 1: cull scan, 32 entries per trip (one mask word)", extra)
    report("blend_backward.hip", "blend_backward", r"v_rsq_f32", "blend_backward: one visited entry per wave and trip (gradient block + wave reduction)", extra)
