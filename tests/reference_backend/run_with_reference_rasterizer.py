#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: run an unchanged reference script exactly as launch/run_reference_script.py does, but with the REFERENCE's
own rasterizer kernels (tests/reference_backend/diff_gaussian_rasterization: oracle/_ref/libgof_cudaref.so) in place of the product's:
the module is imported first, so every later `import diff_gaussian_rasterization` -- the launcher's, gaussian_renderer's -- finds it
in sys.modules.  Everything else (scene loading, simple_knn, the epilogue the environment selects) is the launcher's.
    python tests/reference_backend/run_with_reference_rasterizer.py <reference>/train.py -s <scene> -m <model> ...
Used by tests/test_trajectory_gpu.py with GOF_TORCH_EPILOGUE=1 (the reference's own loss / optimizer / densification on both sides)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import diff_gaussian_rasterization as DGR  # noqa: E402

assert getattr(DGR, "IS_REFERENCE_BACKEND", False), "the product's diff_gaussian_rasterization was imported instead of the reference-kernel one"
os.environ.setdefault("GOF_INTEGRATE_CACHE_GB", "0")        # (the launcher's integrate cache / fused view loop are the product's: not here)
os.environ.setdefault("GOF_TORCH_VIEW_REDUCE", "1")
sys.path.insert(0, os.path.join(ROOT, "gaussian-opacity-fields_amd", "launch"))
import run_reference_script  # noqa: E402

run_reference_script.main()
