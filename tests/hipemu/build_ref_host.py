"""TEST INFRASTRUCTURE.  Compiles the REFERENCE's CUDA rasterizer (/root/reference/submodules/diff-gaussian-rasterization/
cuda_rasterizer/{forward,backward,rasterizer_impl}.cu) as HOST code on top of tests/hipemu (kernels as fibers), from the sources where
they lie, into oracle/_ref/libgof_cudaref_host.so (git-ignored, like the gfx950 builds of oracle/build_ref.sh) behind the same C
wrapper (oracle/ref_capi.cpp).  With it the CPU suite can hold the oracle to the reference's OWN kernels without a GPU
(tests/test_oracle_pins.py); on a GPU box the gfx950 builds do that (tests/test_reference_gpu.py).

Nothing of the reference is copied into the repository.  The sources are transformed on the fly into temporary files that are deleted
after compilation: the CUDA launch chevrons `k << <g, b >> > (args)` become `hipLaunchKernelGGL((k), dim3(g), dim3(b), 0, 0, args)`
(a host compiler cannot parse chevrons) and one brace initialiser clang rejects is normalised, exactly as oracle/build_ref.sh does.
cooperative_groups / cub / cuda_runtime are the small stand-ins of tests/hipemu/ref_shim_host.  -ffp-contract=off: the oracle's
arithmetic contract.

    python tests/hipemu/build_ref_host.py            # no-op when /root/reference is absent"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/submodules/diff-gaussian-rasterization"
OUT = os.path.join(ROOT, "oracle", "_ref")
LIB = os.path.join(OUT, "libgof_cudaref_host.so")
KNN = "/root/reference/submodules/simple-knn"
KNN_LIB = os.path.join(OUT, "libgof_knnref_host.so")
CXX = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")

LAUNCH = re.compile(r"(\b\w+(?:\s*<[^<>;(){}]*>)?)\s*<<\s*<(.*?)>>\s*>\s*\(", re.S)


def _split_top(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    out.append(cur)
    return [x.strip() for x in out]


def _launches_to_calls(text):
    def rep(m):
        cfg = _split_top(m.group(2))
        assert len(cfg) == 2, cfg
        return "hipLaunchKernelGGL((%s), dim3(%s), dim3(%s), 0, nullptr, " % (m.group(1).strip(), cfg[0], cfg[1])
    # (commented-out launches stay comments: the replacement keeps everything on the line it was)
    return LAUNCH.sub(rep, text)


def available():
    return os.path.exists(LIB)


def build(force=False, verbose=False):
    if not os.path.isdir(REF):
        return None
    deps = [os.path.join(REF, "cuda_rasterizer", f) for f in os.listdir(os.path.join(REF, "cuda_rasterizer"))]
    deps += [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "hipemu_rt.cpp"), os.path.join(ROOT, "oracle", "ref_capi.cpp"),
             os.path.join(ROOT, "include", "gof_hip.h"), __file__]
    for d, _, fs in os.walk(os.path.join(HERE, "ref_shim_host")):
        deps += [os.path.join(d, f) for f in fs]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="tmp_host.", dir=OUT)
    try:
        inc = ["-I", os.path.join(HERE, "ref_shim_host"), "-I", os.path.join(HERE, "include"), "-I", os.path.join(REF, "third_party", "glm"),
               "-I", os.path.join(REF, "cuda_rasterizer"), "-I", REF, "-I", os.path.join(ROOT, "oracle")]
        flags = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-w", "-ffp-contract=off", "-mfma", "-mavx2", "-fno-strict-aliasing", "-pthread", "-DGLM_FORCE_CUDA", "-Wno-c++11-narrowing", "-ferror-limit=100"]
        procs, objs = [], []
        for f in ("forward", "backward", "rasterizer_impl"):
            text = open(os.path.join(REF, "cuda_rasterizer", f + ".cu")).read()
            text = text.replace("float2 projected_xy[MAX_NUM_PROJECTED] = { 0.f };", "float2 projected_xy[MAX_NUM_PROJECTED] = {};")
            text = _launches_to_calls(text)
            src = os.path.join(tmp, f + ".cu.cpp")
            open(src, "w").write(text)
            obj = os.path.join(tmp, f + ".o")
            objs.append(obj)
            procs.append((f, subprocess.Popen([CXX] + flags + inc + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for name, src in (("capi", os.path.join(ROOT, "oracle", "ref_capi.cpp")), ("rt", os.path.join(HERE, "hipemu_rt.cpp"))):
            obj = os.path.join(tmp, name + ".o")
            objs.append(obj)
            procs.append((name, subprocess.Popen([CXX] + flags + inc + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        bad = False
        for name, p in procs:
            log = p.communicate()[0]
            if p.returncode != 0:
                bad = True
                sys.stderr.write("---- %s ----\n%s\n" % (name, log[-5000:]))
        if bad:
            raise RuntimeError("build_ref_host: compilation failed")
        subprocess.check_call([CXX, "-shared", "-pthread", "-o", LIB] + objs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


def build_knn(force=False):
    """the reference's simple-knn (submodules/simple-knn/simple_knn.cu: Morton sort + 1024-point boxes + exact 3-NN), same treatment,
    behind oracle/ref_knn_capi.cpp"""
    if not os.path.isdir(KNN):
        return None
    deps = [os.path.join(KNN, "simple_knn.cu"), os.path.join(KNN, "simple_knn.h"), os.path.join(ROOT, "oracle", "ref_knn_capi.cpp"),
            os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "hipemu_rt.cpp"), __file__]
    for d, _, fs in os.walk(os.path.join(HERE, "ref_shim_host")):
        deps += [os.path.join(d, f) for f in fs]
    if not force and os.path.exists(KNN_LIB) and all(os.path.getmtime(KNN_LIB) >= os.path.getmtime(d) for d in deps):
        return KNN_LIB
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="tmp_host.", dir=OUT)
    try:
        inc = ["-I", os.path.join(HERE, "ref_shim_host"), "-I", os.path.join(HERE, "include"), "-I", KNN]
        flags = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-w", "-ffp-contract=off", "-mfma", "-mavx2", "-fno-strict-aliasing", "-pthread",
                 "-Wno-c++11-narrowing", "-include", "cfloat"]
        src = os.path.join(tmp, "simple_knn.cu.cpp")
        open(src, "w").write(_launches_to_calls(open(os.path.join(KNN, "simple_knn.cu")).read()))
        objs = []
        for name, path in (("knn", src), ("capi", os.path.join(ROOT, "oracle", "ref_knn_capi.cpp")), ("rt", os.path.join(HERE, "hipemu_rt.cpp"))):
            obj = os.path.join(tmp, name + ".o")
            subprocess.check_call([CXX] + flags + inc + ["-c", path, "-o", obj])
            objs.append(obj)
        subprocess.check_call([CXX, "-shared", "-pthread", "-o", KNN_LIB] + objs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return KNN_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_knn(force="--force" in sys.argv))
