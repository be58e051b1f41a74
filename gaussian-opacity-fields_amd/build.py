#!/usr/bin/env python3
"""Build libgof_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: a few translation units
compiled in parallel and linked into gaussian-opacity-fields_amd/lib/libgof_hip.so."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# GOF_BUILD_TAG=<tag> builds an experimental variant next to the product (build_<tag>/, lib/libgof_hip_<tag>.so; select it at run
# time with GOF_HIP_LIB) -- used to time compiler-flag / macro variants in one GPU session
_TAG = os.environ.get("GOF_BUILD_TAG", "")
OBJ = os.path.join(HERE, "build" + ("_" + _TAG if _TAG else ""))
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgof_hip%s.so" % ("_" + _TAG if _TAG else ""))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the arithmetic contract -- fused multiply-adds only where fmaf() is written
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result",
         # the LLVM "atomic optimizer" rewrites few-lane same-address LDS/global atomics into a per-lane
         # readlane loop (17 loops per splat in blend_backward); the hardware handles them directly
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         # the SLP vectorizer pairs fp32 ops into v_pk_mul/add_f32 and pays for it with v_mov shuffles (50 of 232 instructions in
         # the backward's hot block) and 12 more VGPRs (occupancy 4 instead of 5): measured -13 % blend_backward, -6 % blend_forward
         "-fno-slp-vectorize"] + os.environ.get("GOF_EXTRA_FLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    """Every header a translation unit may include: csrc/*.h and ALL of include/*.h (gof_train_hip.h / gof_knn_hip.h carry ABI structs)."""
    inc = os.path.join(HERE, "..", "include")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]


def flags_stamp():
    """The flag set is part of an object's identity: a stamp file in the object directory records it, a change rebuilds everything."""
    import hashlib
    stamp = os.path.join(OBJ, "flags.sha")
    want = hashlib.sha256(" ".join([HIPCC] + FLAGS).encode()).hexdigest()
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if have != want:
        for f in os.listdir(OBJ):
            if f.endswith(".o"):
                os.remove(os.path.join(OBJ, f))
        with open(stamp, "w") as fh:
            fh.write(want)


def needs(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def compile_one(src, extra=()):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace(".hip", ".o"))
    if needs(o, [s] + headers()):
        cmd = [HIPCC] + FLAGS + list(extra) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return o


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    flags_stamp()
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if needs(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
