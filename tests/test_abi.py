"""CPU-only: the C-ABI library loads and exports every symbol include/*.h declares;
host-only size queries behave."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gof_hip.h")
LIB = os.path.join(ROOT, "gaussian-opacity-fields_amd", "lib", "libgof_hip.so")


def declared_functions():
    inc = os.path.join(ROOT, "include")
    src = "\n".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gof_[a-z0-9_]+)\s*\(", src)))


def test_library_exists_and_loads():
    assert os.path.exists(LIB), "build with python gaussian-opacity-fields_amd/build.py"
    ctypes.CDLL(LIB)


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(LIB)
    names = declared_functions()
    assert len(names) >= 22 and "gof_adam_step" in names and "gof_forward_render" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_size_queries_are_host_only_and_monotone():
    from diff_gaussian_rasterization import _backend as B
    L = B.lib
    assert L.gof_abi_version() >= 5
    assert L.gof_geom_bytes(0) > 0
    assert L.gof_geom_bytes(1000) < L.gof_geom_bytes(100000)
    assert L.gof_geom_bytes(1_000_000) >= 1_000_000 * (64 + 16 + 4 + 4 + 4 + 1)
    # the forward / backward pair's geometry workspace: the same layout without its tail, the query's 16 B per Gaussian (ABI 12)
    assert 0 < L.gof_geom_bytes(1_000_000) - L.gof_geom_bytes_forward(1_000_000) - 16_000_000 < 4096
    assert L.gof_image_bytes(1600, 1063) >= 1600 * 1063 * 24
    assert L.gof_binning_bytes(0, 400, 400) > 0
    assert L.gof_binning_bytes(5_000_000, 1600, 1063) >= 5_000_000 * 16
    assert L.gof_point_bytes(10) < L.gof_point_bytes(10_000_000)


def test_struct_layout_matches_header():
    """The ctypes mirrors (product binding and oracle binding) agree with each other and with the
    field list of the header."""
    from diff_gaussian_rasterization import _backend as B
    import oracle_binding as ob
    src = open(HEADER).read()
    body = src[src.index("typedef struct GofRasterArgs {"):src.index("} GofRasterArgs;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split(",")
        first = names[0].split()[-1]
        fields.append(first)
        fields.extend(n.strip() for n in names[1:])
    assert [f[0] for f in B.GofRasterArgs._fields_] == fields
    assert [f[0] for f in ob.GofRasterArgs._fields_] == fields
    assert ctypes.sizeof(B.GofRasterArgs) == ctypes.sizeof(ob.GofRasterArgs) == 11 * 4 + 4 + 14 * 8 + 4 * 4


def test_train_epilogue_struct_and_host_queries():
    """GofAdamTensor mirror matches include/gof_train_hip.h; host-only queries and argument checks need no GPU."""
    from train_epilogue import _backend as TB
    src = open(os.path.join(ROOT, "include", "gof_train_hip.h")).read()
    body = src[src.index("typedef struct GofAdamTensor {"):src.index("} GofAdamTensor;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [d.strip().replace("*", " ").split()[-1] for d in body.split("{", 1)[1].split(";") if d.strip()]
    assert [f[0] for f in TB.GofAdamTensor._fields_] == fields
    assert ctypes.sizeof(TB.GofAdamTensor) == 4 * 8 + 8 + 2 * 4
    L = TB.lib
    assert L.gof_ssim_scratch_bytes(3, 1600, 1063) >= 3 * 100 * 67 * 4
    assert L.gof_adam_step(0, None, 0.9, 0.999, 1e-15, None) == 0
    assert L.gof_adam_step(17, None, 0.9, 0.999, 1e-15, None) < 0 and b"n_tensors" in L.gof_last_error()
    assert L.gof_ssim_forward(0, 8, 8, None, None, None, None, None, None, 0, None) < 0
    assert L.gof_depth_to_normal(8, 8, None, None, 1.0, 1.0, None, None, None) < 0
    assert len(TB.window_taps()) == 11 and abs(sum(TB.window_taps()) - 1.0) < 1e-6


def test_headers_are_plain_c():
    """The boundary is a C ABI: every header under include/ must compile as C99 on its own (no C++ constructs, no torch / HIP types),
    and together (no clashing declarations)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    inc = os.path.join(ROOT, "include")
    headers = sorted(glob.glob(os.path.join(inc, "*.h")))
    assert len(headers) >= 3
    for h in headers:
        r = subprocess.run([gcc, "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-pedantic", "-Werror", "-x", "c", h], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "all.c")
        with open(src, "w") as f:
            f.write("".join('#include "%s"\n' % os.path.basename(h) for h in headers) + "int main(void) { return gof_abi_version() > 0 ? 0 : 1; }\n")
        r = subprocess.run([gcc, "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-I", inc, src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_a_c_client_links_against_the_library():
    """What the reference's maintainer would do from C / cgo / JNI: include the headers, link libgof_hip.so, call it.  Host-only
    entry points (workspace sizes, ABI version, argument validation) run without a GPU."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    lib_dir = os.path.join(ROOT, "gaussian-opacity-fields_amd", "lib")
    code = r"""
#include <stdio.h>
#include "gof_hip.h"
#include "gof_train_hip.h"
#include "gof_knn_hip.h"
int main(void) {
    GofRasterArgs a = {0};
    size_t g1 = gof_geom_bytes(1000), g2 = gof_geom_bytes(2000);
    if (!(g2 > g1 && gof_image_bytes(1600, 1063) > 0 && gof_binning_bytes(1000000u, 1600, 1063) > 0)) return 2;
    if (gof_abi_version() < 3) return 3;
    a.P = 10; a.W = 0; a.H = 16;                                   /* invalid: reported, not crashed */
    unsigned int n = 0;
    if (gof_forward_prepare(&a, (void*)0, 0, (void*)0, 0, (int*)0, &n, (void*)0) >= 0) return 4;
    if (gof_last_error()[0] == 0) return 5;
    if (gof_mtets_tet_ws_bytes(1000000) >= gof_mtets_tet_ws_bytes(1000000) + gof_mtets_edge_ws_bytes(1000000)) return 6;
    printf("abi %d geom %zu\n", gof_abi_version(), g1);
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "client.c"), os.path.join(d, "client")
        with open(src, "w") as f:
            f.write(code)
        r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", lib_dir, "-lgof_hip",
                            "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([exe], capture_output=True, text=True, env={**os.environ, "LD_LIBRARY_PATH": lib_dir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", "")})
        assert r.returncode == 0 and r.stdout.startswith("abi "), (r.returncode, r.stdout, r.stderr)
