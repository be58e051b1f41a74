"""Build tests/hipemu/_build/libgof_hip_emu.so: the product's kernel sources (gaussian-opacity-fields_amd/csrc/*.hip) compiled as HOST
C++ against tests/hipemu/include/hip/hip_runtime.h (fibers in wave64 lock step).  Test infrastructure: see that header.

    python tests/hipemu/build_emu.py [--asan] [--force]

The sources are compiled where they lie; the one construct a host compiler cannot take (`extern __shared__ T name[];`, dynamic LDS)
is rewritten in a scratch copy.  -ffp-contract=off as in the product build, -mfma so that fmaf() is the hardware instruction."""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "gaussian-opacity-fields_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
DYN_LDS = re.compile(r"extern\s+__shared__\s+(\w+)\s+(\w+)\s*\[\s*\]\s*;")

# Places where a kernel relies on the wave's instruction-level lock step without a cross-lane operation (fibers run a lane up to its
# next synchronisation point, so "all lanes read a word, then one lane writes it" needs an explicit wave barrier in the emulation).
# (file, text as it stands in the source, replacement).  Every entry must match at least once, or the build fails: a stale entry
# cannot go unnoticed.
LOCKSTEP_POINTS = [
    # radix.hip, rs_scatter / os_pass: the lanes of a digit group read the group's cursor, then its lowest lane advances it
    ("radix.hip", "if (cnt[s]) s_cur[wave][d] = base + cnt[s];", "HIPEMU_WAVE_SYNC(); if (cnt[s]) s_cur[wave][d] = base + cnt[s]; HIPEMU_WAVE_SYNC();"),
    # binning.hip, order_tiles / place: the same pattern on the wave's running rank base of a cost class (inside `if (live)`: the
    # read is hoisted in front of the branch so that every lane of the wave reaches the barrier)
    ("binning.hip", "        if (live) {\n            const uint32_t base = s_wc[wave][bkt]; ",
     "        const uint32_t base_ = live ? s_wc[wave][bkt] : 0u; HIPEMU_WAVE_SYNC();\n        if (live) {\n            const uint32_t base = base_; "),
    ("binning.hip", "            order[x * stride + s_off[x][bkt] + (r - lo)] = i;\n        }\n", "            order[x * stride + s_off[x][bkt] + (r - lo)] = i;\n        }\n        HIPEMU_WAVE_SYNC();\n"),
]


def lib_path(tag=""):
    return os.path.join(OUT, "libgof_hip_emu%s.so" % (("_" + tag) if tag else ""))


def build(asan=False, force=False, extra_flags=(), tag="", verbose=False):
    """One builder at a time per output (an exclusive file lock): pytest-xdist workers that find the stamp stale would otherwise compile
    into the same object files and link / load a half-written library (seen as one spuriously failing test per `-n 8` run after a
    kernel edit)."""
    import fcntl
    os.makedirs(os.path.join(OUT, "obj" + tag), exist_ok=True)
    with open(os.path.join(OUT, "lock%s_%s" % (tag, "asan" if asan else "plain")), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(asan, force, extra_flags, tag, verbose)


def _build_locked(asan, force, extra_flags, tag, verbose):
    flags = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-mfma", "-mavx2", "-fno-strict-aliasing", "-pthread",
             "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-unused-value", "-Wno-pass-failed",
             "-I", os.path.join(HERE, "include"), "-I", CSRC, "-DGOF_HIPEMU=1"] + list(extra_flags)
    if asan:
        flags += ["-fsanitize=address,undefined", "-fno-sanitize=pointer-overflow", "-fno-omit-frame-pointer"]      # (the layout functions size a workspace by carving from a null base)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "hipemu_rt.cpp"),
                                                                os.path.join(ROOT, "include", "gof_hip.h"), __file__]
    h = hashlib.sha256((" ".join(flags)).encode())
    for d in sorted(deps):
        h.update(open(d, "rb").read())
    stamp = os.path.join(OUT, "stamp%s_%s" % (tag, "asan" if asan else "plain"))
    out = lib_path(tag + ("asan" if asan else ""))
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return out
    objs = []
    procs = []
    for f in srcs + ["hipemu_rt.cpp"]:
        src = os.path.join(CSRC, f) if f.endswith(".hip") else os.path.join(HERE, f)
        text = open(src).read()
        orig = text
        text = DYN_LDS.sub(r"\1* \2 = reinterpret_cast<\1*>(hipemu::dynamic_lds());", text)
        for fname, old, new in LOCKSTEP_POINTS:
            if fname == f:
                if old not in text:
                    raise RuntimeError("hipemu: lock-step point not found in %s: %r" % (f, old))
                text = text.replace(old, new)
        if text != orig:
            text = '#line 1 "%s"\n' % src + text
            src = os.path.join(OUT, "obj" + tag, f + ".cpp")
            open(src, "w").write(text)
        obj = os.path.join(OUT, "obj" + tag, f + (".asan" if asan else "") + ".o")
        objs.append(obj)
        cmd = [CXX] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for f, p in procs:
        log = p.communicate()[0]
        if p.returncode != 0:
            failed = True
            sys.stderr.write("---- %s ----\n%s\n" % (f, log[-6000:]))
        elif verbose and log.strip():
            sys.stderr.write("---- %s (warnings) ----\n%s\n" % (f, log[-1500:]))
    if failed:
        raise RuntimeError("hipemu build failed")
    link = [CXX, "-shared", "-pthread", "-o", out + ".tmp"] + objs + (["-fsanitize=address,undefined"] if asan else [])
    subprocess.check_call(link)
    os.replace(out + ".tmp", out)          # a process that already maps the old library keeps its inode
    open(stamp, "w").write(h.hexdigest())
    return out


if __name__ == "__main__":
    print(build(asan="--asan" in sys.argv, force="--force" in sys.argv, verbose=True))
