#!/usr/bin/env python3
"""Run an UNCHANGED script of the reference (train.py, render.py, extract_mesh.py, ...) against the
MI355X backend.

    python gaussian-opacity-fields_amd/launch/run_reference_script.py /path/to/gaussian-opacity-fields/train.py -s <scene> ...

What it does, in this order (nothing in the reference checkout is edited):
  1. puts this package first on sys.path so `import diff_gaussian_rasterization` resolves to the gfx950
     backend, and the reference checkout second so its own packages (scene, utils, gaussian_renderer) import;
  2. puts the import shims for the out-of-scope native submodules (simple_knn, tetranerf) on sys.path unless the
     real modules are importable;
  3. rebinds utils.tetmesh.marching_tetrahedra to the HIP implementation BEFORE the script imports it by name
     (extract_mesh.py:14: `from utils.tetmesh import marching_tetrahedra`);
  4. runs the script with runpy as __main__ with the remaining argv.
"""
import importlib
import os
import runpy
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        sys.exit(2)
    script = os.path.abspath(sys.argv[1])
    ref_root = os.path.dirname(script)
    sys.path.insert(0, ref_root)
    sys.path.insert(0, PKG)
    for mod in ("simple_knn._C", "tetranerf.utils.extension"):
        try:
            importlib.import_module(mod)
        except Exception:
            shim = os.path.join(PKG, "shims")
            if shim not in sys.path:
                sys.path.append(shim)
    import diff_gaussian_rasterization  # noqa: F401  fail early and loudly if libgof_hip.so is missing
    try:
        import utils.tetmesh as ref_tetmesh
        import tetmesh as hip_tetmesh
        ref_tetmesh.marching_tetrahedra = hip_tetmesh.marching_tetrahedra
    except ImportError:
        pass
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
