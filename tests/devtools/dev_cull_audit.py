"""Audit of the forward blend's cull scan (library built with -DGOF_STATS -DGOF_CULL_AUDIT, lib/libgof_hip_audit.so, selected with
GOF_HIP_LIB): the consumption walks EVERY list entry and counts the (pixel, entry) pairs the exact path accepts (t > 0.2 and
alpha >= 1/255, before the pixel saturates) that the footprint-conic scan had not marked as candidates.  Prints one JSON line per
scene: {"scene", "accepted_pairs", "candidates", "dropped_by_the_scan"} -- the last must be 0 -- and, since round 5, the same two counts
for the ray-centric pixel pass of the opacity-field query ("integrate_accepted_pairs", "integrate_dropped_by_the_scan").
    GOF_HIP_LIB=.../libgof_hip_audit.so python tests/devtools/dev_cull_audit.py s1m stress_box ..."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gpu_common import product_forward_raw, to_dev  # noqa: E402
import synthetic_scenes as S  # noqa: E402
from diff_gaussian_rasterization import _backend as B  # noqa: E402


def scenes():
    from test_parity_gpu import SCENES
    table = dict(SCENES)
    table["s1m"] = lambda: S.scene_frustum(1_000_000, seed=0)
    table["s1m_ks01"] = lambda: S.scene_frustum(1_000_000, seed=0, kernel_size=0.1)
    table["s1m_posed"] = lambda: S.scene_frustum(1_000_000, seed=0, pose_seed=0)
    table["far_subpixel_posed"] = lambda: S.scene_frustum(300_000, W=800, H=528, focal=600.0, seed=7, sigma_px=0.4, zmin=5.0, zmax=80.0, pose_seed=9)
    table["far_subpixel"] = lambda: S.scene_frustum(300_000, W=800, H=528, focal=600.0, seed=7, sigma_px=0.4, zmin=5.0, zmax=80.0)
    return table


if __name__ == "__main__":
    assert hasattr(B.lib, "gof_debug_fw_stats"), "this needs the audit build of the library (GOF_HIP_LIB=.../libgof_hip_audit.so)"
    out = (C.c_ulonglong * 8)()
    table = scenes()
    for name in sys.argv[1:]:
        sd = to_dev(table[name]())
        B.lib.gof_debug_fw_stats(out, 1)
        product_forward_raw(sd)
        torch.cuda.synchronize()
        B.lib.gof_debug_fw_stats(out, 1)
        s = list(out)
        row = {"scene": name, "accepted_pairs": s[3], "candidates": s[1], "dropped_by_the_scan": s[6]}
        # round 5: the same audit of the opacity-field query's ray-centric pixel pass (integrate_rays: one ray per lane, the conic at the
        # lane's own ray): [7] (ray, entry) pairs accepted, [11] of them outside the scan's candidates
        if hasattr(B.lib, "gof_debug_int_stats"):
            from diff_gaussian_rasterization import GaussianRasterizer
            from gpu_common import settings_from
            iout = (C.c_ulonglong * 16)()
            B.lib.gof_debug_int_stats(iout, 1)
            pts = sd["means3D"][:1000].contiguous()
            GaussianRasterizer(settings_from(sd)).integrate(points3D=pts, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"],
                                                            scales=sd["scales"], rotations=sd["rotations"])
            torch.cuda.synchronize()
            B.lib.gof_debug_int_stats(iout, 1)
            row["integrate_accepted_pairs"], row["integrate_dropped_by_the_scan"] = iout[7], iout[11]
        print(json.dumps(row), flush=True)
