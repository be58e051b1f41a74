"""Is the DEFAULT-build gfx950 code of the two blend kernels at the working tree the same, instruction for instruction, as at a git
revision?  (An edit under a developer-only #ifdef -- GOF_STATS, GOF_TILE_CLOCK -- changes the source hash bench.py labels the committed
PMC pass with, not the kernels the pass measured.)  Compiles both states of csrc/blend_{forward,backward}.hip with the product's flags
to device assembly and compares it, ignoring the per-source `__hip_cuid_*` symbol, file names and line tables.
    python tests/devtools/dev_same_isa.py <git-rev> [--record]     # --record: add the current source hash to the PMC file's _same_isa_sha16
No GPU needed (hipcc cross-compiles)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "gaussian-opacity-fields_amd", "csrc")
sys.path.insert(0, ROOT)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         "-fno-slp-vectorize", "-S", "--cuda-device-only"]
FILES = ["blend_forward.hip", "blend_backward.hip"]


def asm_of(path_in_csrc):
    out = tempfile.mktemp(suffix=".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, path_in_csrc], stderr=subprocess.DEVNULL)
    text = open(out).read(); os.remove(out)
    keep = [re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid", ln) for ln in text.splitlines() if not re.match(r"\s*\.(file|loc|ident)\b", ln)]
    return [ln for ln in keep if "_same_isa_tmp" not in ln]


def main():
    rev = sys.argv[1]
    same = True
    same_by = {}
    for f in FILES:
        cur = asm_of(os.path.join(CSRC, f))
        tmp = os.path.join(CSRC, "_same_isa_tmp_" + f)          # (inside csrc so that the relative includes resolve; headers: working tree)
        try:
            open(tmp, "w").write(subprocess.check_output(["git", "-C", ROOT, "show", "%s:gaussian-opacity-fields_amd/csrc/%s" % (rev, f)], text=True))
            old = asm_of(tmp)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        ok = cur == old
        same &= ok
        same_by[f[:-4]] = ok
        print("%-22s %s (%d lines of device assembly)" % (f, "identical" if ok else "DIFFERENT", len(cur)))
    hdr = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", rev, "--", "gaussian-opacity-fields_amd/csrc/gof_common.h"]).returncode == 0
    print("gof_common.h           %s since %s" % ("unchanged" if hdr else "CHANGED (the comparison above used the working tree's header for both states)", rev))
    if hdr and "--record" in sys.argv:
        import bench
        f = os.path.join(ROOT, "profiles", bench.PMC_FILE)
        d = json.load(open(f))
        for k, ok in same_by.items():            # per kernel: the bench line quotes the dominant kernel's counters only
            cur = bench.kernel_sha16(k)
            known = [d.get("_sha16_by_kernel", {}).get(k)] + d.get("_same_isa_sha16_by_kernel", {}).get(k, [])
            if ok and cur not in known:
                d.setdefault("_same_isa_sha16_by_kernel", {}).setdefault(k, []).append(cur)
                print("recorded", k, cur, "in", f)
        json.dump(d, open(f, "w"), indent=1)
    sys.exit(0 if (same and hdr) else 1)


if __name__ == "__main__":
    main()
