#!/usr/bin/env python3
"""Run an UNCHANGED script of the reference (train.py, render.py, extract_mesh.py, ...) against the
MI355X backend.

    python gaussian-opacity-fields_amd/launch/run_reference_script.py /path/to/gaussian-opacity-fields/train.py -s <scene> ...

What it does, in this order (nothing in the reference checkout is edited):
  1. puts this package first on sys.path so `import diff_gaussian_rasterization` resolves to the gfx950
     backend, and the reference checkout second so its own packages (scene, utils, gaussian_renderer) import;
  2. `simple_knn._C.distCUDA2` resolves to the HIP implementation in this package (simple_knn/, include/gof_knn_hip.h); puts the
     import shim for the out-of-scope native submodule (tetranerf: CGAL Delaunay) on sys.path unless the
     real modules are importable;
  3. rebinds utils.tetmesh.marching_tetrahedra to the HIP implementation BEFORE the script imports it by name
     (extract_mesh.py:14: `from utils.tetmesh import marching_tetrahedra`);
  4. rebinds the per-iteration training epilogue to its HIP implementation (train_epilogue/, include/gof_train_hip.h):
     utils.loss_utils.ssim (train.py:20), utils.depth_utils.depth_to_normal / depths_to_points (train.py:38) and the
     optimizer GaussianModel.training_setup builds (scene/gaussian_model.py:360 -> FusedAdam over the same param groups) and
     GaussianModel.compute_3D_filter (scene/gaussian_model.py:262-311, one launch over points x cameras) and
     GaussianModel.densify_and_prune (:685-707: device index lists + one row gather per tensor; GOF_TORCH_DENSIFY=1 keeps the reference's).
     GOF_TORCH_EPILOGUE=1 keeps the reference's torch implementations; the loss train.py composes inline from these helpers
     (train.py:150-189) is evaluated by ONE fused call at `loss.backward()` (train_epilogue/deferred.py; GOF_EAGER_LOSS=1 keeps the
     helpers eager, one launch pair each); scene.cameras.Camera.world_view_transform becomes a
     train_epilogue.PoseMatrix (train.py:177-179: `.T.inverse()` computed once per camera, `c2w[:3, :3] @ normals` one streaming
     launch instead of a GEMM; GOF_PLAIN_POSE=1 keeps the plain tensor);
  5. wraps gaussian_renderer.integrate (imported by name at extract_mesh.py:5) so that the Gaussian side of the opacity-field
     query (binning + pixel pass) runs once per view for the ~10 point sets extract_mesh.py queries against the unchanged model
     (diff_gaussian_rasterization.integrate_view_key; GOF_INTEGRATE_CACHE_GB bounds the HBM it may keep, 0 disables);
  6. runs the script as __main__ with the remaining argv.  A function the script defines itself cannot be rebound through an
     imported module: for extract_mesh.py's `evaluage_alpha` (:17-34, the loop over all views per point set) the module body is
     executed without its `if __name__ == "__main__":` block, the name is pointed at mesh_extraction.evaluate_alpha (the same loop
     with the min / arg-min reduction over views fused into the point pass), then the block runs (run_script below;
     GOF_TORCH_VIEW_REDUCE=1 keeps the script's own function).
"""
import importlib
import os
import runpy
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rebind_train_epilogue():
    """Swap the reference's pure-torch loss helpers and optimizer for the HIP ones, by name, before the script's
    `from ... import` statements run.  Modules that are not importable (a script that does not train) are skipped."""
    import train_epilogue as T
    from train_epilogue import deferred
    # train.py:150-189 composes its loss inline from these helpers, the image's channel slices and the camera pose: with deferred
    # evaluation on, the script's own lines collect coefficients and loss.backward() is ONE fused call (train_epilogue/deferred.py: any
    # other spelling computes eagerly with the mirrors below); GOF_EAGER_LOSS=1 keeps every line eager
    deferred.enable(os.environ.get("GOF_EAGER_LOSS") != "1")
    try:
        import utils.loss_utils as ref_loss
        ref_loss.ssim = deferred.ssim                       # (eager: T.ssim)
        ref_loss.l1_loss = deferred.l1_loss                 # (eager: T.l1_loss)
    except ImportError:
        pass
    try:
        import utils.depth_utils as ref_depth
        ref_depth.depth_to_normal = deferred.depth_to_normal      # (eager: T.depth_to_normal)
        ref_depth.depths_to_points = T.depths_to_points
    except ImportError:
        pass
    # train.py:177-179 inverts the camera pose and multiplies its 3x3 block with the normal image every iteration: the Camera's
    # world_view_transform becomes a tensor that remembers its inverse and applies the block with one streaming launch
    # (train_epilogue/pose.py); GOF_PLAIN_POSE=1 keeps the plain tensor
    if os.environ.get("GOF_PLAIN_POSE") != "1":
        try:
            import scene.cameras as ref_cameras
            _cam_init = ref_cameras.Camera.__init__

            def _camera_init(self, *a, **k):
                _cam_init(self, *a, **k)
                self.world_view_transform = T.PoseMatrix.wrap(self.world_view_transform)
            ref_cameras.Camera.__init__ = _camera_init
        except Exception:
            pass
    try:
        from scene.gaussian_model import GaussianModel
    except Exception:
        return
    _setup = GaussianModel.training_setup

    def training_setup(self, training_args):
        _setup(self, training_args)
        groups = self.optimizer.param_groups               # six named groups with their lr (gaussian_model.py:349-358)
        old = self.optimizer
        self.optimizer = T.FusedAdam([{"params": g["params"], "lr": g["lr"], "name": g["name"]} for g in groups], lr=0.0, eps=1e-15)
        for hook in getattr(old, "_optimizer_step_pre_hooks", {}).values():      # e.g. the DP gradient all-reduce (run_train_dp.py)
            self.optimizer.register_step_pre_hook(hook)
        for hook in getattr(old, "_optimizer_step_post_hooks", {}).values():
            self.optimizer.register_step_post_hook(hook)
    GaussianModel.training_setup = training_setup
    GaussianModel.compute_3D_filter = T.compute_3D_filter      # train.py:118,261,269: after every densification
    GaussianModel.add_densification_stats = T.add_densification_stats   # train.py:256: every iteration until densify_until_iter
    # train.py:260: clone / split / prune as ordered device index lists + one row gather per tensor (train_epilogue/densify.py)
    if os.environ.get("GOF_TORCH_DENSIFY") != "1":
        current = GaussianModel.densify_and_prune
        if hasattr(current, "inner"):          # the data-parallel launcher's wrapper (statistics all-reduce first): replace what it calls
            current.inner = T.densify_and_prune
        else:
            GaussianModel.densify_and_prune = T.densify_and_prune
    # the three derived tensors render() reads every iteration (gaussian_renderer/__init__.py:60,70-71)
    GaussianModel.get_scaling_with_3D_filter = property(T.activations.get_scaling_with_3D_filter)
    GaussianModel.get_opacity_with_3D_filter = property(T.activations.get_opacity_with_3D_filter)
    GaussianModel.get_rotation = property(T.activations.get_rotation)
    # get_features (gaussian_model.py:173-176) concatenates _features_dc and _features_rest -- 192 B per Gaussian copied, and the
    # gradient split back, every iteration.  render() / integrate() hand the result straight to the rasterizer
    # (gaussian_renderer/__init__.py:94,194), which reads the two stored tensors directly when given a SplitSH; any other use of
    # the object (pipe.convert_SHs_python, :84-85) sees the concatenation.  GOF_CAT_FEATURES=1 keeps the reference's property.
    if os.environ.get("GOF_CAT_FEATURES") != "1":
        from diff_gaussian_rasterization import SplitSH
        GaussianModel.get_features = property(lambda self: SplitSH(self._features_dc, self._features_rest))


def rebind_integrate_with_view_cache():
    """extract_mesh.py:23-31 calls integrate(points, view, gaussians, ...) for every view, once per bisection step, with the
    same Gaussians.  The wrapper announces a key made of the identities + version counters of everything but the points."""
    import torch
    import diff_gaussian_rasterization as DGR
    try:
        import gaussian_renderer as GR
    except Exception:
        return
    orig = GR.integrate

    def stamp(t):
        return (t.data_ptr(), t._version, tuple(t.shape)) if isinstance(t, torch.Tensor) else t

    def integrate(points3D, viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier=1.0, override_color=None, subpixel_offset=None):
        if override_color is not None or subpixel_offset is not None or torch.is_grad_enabled():
            return orig(points3D, viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, override_color, subpixel_offset)
        cam = viewpoint_camera
        key = (id(cam), getattr(cam, "uid", None), stamp(cam.world_view_transform), stamp(cam.full_proj_transform),
               int(cam.image_width), int(cam.image_height), float(cam.FoVx), float(cam.FoVy), id(pc), int(pc.active_sh_degree),
               tuple(stamp(getattr(pc, n, None)) for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "filter_3D")),
               stamp(bg_color), float(kernel_size), float(scaling_modifier),
               bool(getattr(pipe, "compute_cov3D_python", False)), bool(getattr(pipe, "convert_SHs_python", False)), bool(getattr(pipe, "debug", False)))
        with DGR.integrate_view_key(key):
            return orig(points3D, viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, override_color, subpixel_offset)
    GR.integrate = integrate


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        sys.exit(2)
    script = os.path.abspath(sys.argv[1])
    ref_root = os.path.dirname(script)
    sys.path.insert(0, ref_root)
    sys.path.insert(0, PKG)
    for mod in ("tetranerf.utils.extension",):
        try:
            importlib.import_module(mod)
        except Exception:
            shim = os.path.join(PKG, "shims")
            if shim not in sys.path:
                sys.path.append(shim)
    import diff_gaussian_rasterization  # noqa: F401  fail early and loudly if libgof_hip.so is missing
    if os.environ.get("GOF_STATS_JSON"):
        # the binding's counters of the whole run (frames redone for a pool / capacity learnt too small, read-backs, shapes that inherited
        # their pools across a densification: _backend._stats) written at exit -- evidence runs read them (tests/devtools/dev_r6_trajectory.py)
        import atexit
        import json

        def _dump_stats(path=os.environ["GOF_STATS_JSON"]):
            st = getattr(getattr(diff_gaussian_rasterization, "_C", None), "_stats", None)
            if st is not None:
                st = dict(st)
                d = sys.modules.get("train_epilogue.deferred")
                if d is not None:                       # how many iterations' losses were one fused call / fell back to the eager mirrors
                    st["deferred_loss"] = dict(d.stats)
                with open(path, "w") as f:
                    json.dump(st, f)
        atexit.register(_dump_stats)
    try:
        import utils.tetmesh as ref_tetmesh
        import tetmesh as hip_tetmesh
        ref_tetmesh.marching_tetrahedra = hip_tetmesh.marching_tetrahedra
    except ImportError:
        pass
    if os.environ.get("GOF_TORCH_EPILOGUE", "0") != "1":
        rebind_train_epilogue()
    if os.environ.get("GOF_INTEGRATE_CACHE_GB", "") not in ("0", "0.0"):
        rebind_integrate_with_view_cache()
    sys.argv = [script] + sys.argv[2:]
    rebind = {}
    if os.environ.get("GOF_TORCH_VIEW_REDUCE", "0") != "1":
        # extract_mesh.py:17-34: the script's own view loop -> the one whose min / arg-min reduction is fused into the point pass
        import mesh_extraction
        rebind["evaluage_alpha"] = lambda points, views, gaussians, pipeline, background, kernel_size, return_color=False: \
            mesh_extraction.evaluate_alpha(points, views, gaussians, pipeline, background, kernel_size, return_color)
    run_script(script, rebind)


def _is_main_guard(node):
    """`if __name__ == "__main__":` at module level"""
    import ast
    if not isinstance(node, ast.If) or not isinstance(node.test, ast.Compare) or len(node.test.ops) != 1:
        return False
    t = node.test
    sides = [t.left, t.comparators[0]]
    return (isinstance(t.ops[0], ast.Eq) and any(isinstance(x, ast.Name) and x.id == "__name__" for x in sides)
            and any(isinstance(x, ast.Constant) and x.value == "__main__" for x in sides))


def run_script(script, rebind):
    """Run `script` as __main__.  Functions the script DEFINES ITSELF (extract_mesh.py's evaluage_alpha) cannot be rebound through an
    imported module: when the script defines one of the names in `rebind`, its module body is executed first WITHOUT the
    `if __name__ == "__main__":` block(s), the names are replaced in its namespace, then the guarded block(s) run -- the script's
    text is untouched and every statement runs exactly once, in order (the guard blocks are at the end of the reference's scripts)."""
    import ast
    import types
    with open(script, "rb") as f:
        src = f.read()
    tree = ast.parse(src, filename=script)
    defined = {n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))}
    names = [k for k in rebind if k in defined]
    guards = [n for n in tree.body if _is_main_guard(n)]
    # only split when every guard block comes after everything else (otherwise the order of execution would change)
    tail_ok = bool(guards) and all(_is_main_guard(n) for n in tree.body[len(tree.body) - len(guards):])
    if not names or not tail_ok:
        runpy.run_path(script, run_name="__main__")
        return
    mod = types.ModuleType("__main__")
    mod.__file__ = script
    mod.__builtins__ = __builtins__
    prev_main = sys.modules.get("__main__")
    sys.modules["__main__"] = mod
    try:
        head = ast.Module(body=[n for n in tree.body if not _is_main_guard(n)], type_ignores=[])
        exec(compile(head, script, "exec"), mod.__dict__)
        for k in names:
            mod.__dict__[k] = rebind[k]
        exec(compile(ast.Module(body=guards, type_ignores=[]), script, "exec"), mod.__dict__)
    finally:
        if prev_main is not None:
            sys.modules["__main__"] = prev_main


if __name__ == "__main__":
    main()
