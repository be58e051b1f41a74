"""Developer script (not a test): first contact of the kernels with the oracle on a GPU."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S, oracle_binding as ob

def run(P, W, H, focal, seed=0, ks=0.0):
    sc = S.scene_frustum(P, W=W, H=H, focal=focal, seed=seed, kernel_size=ks)
    o = ob.OracleScene(sc)
    t0 = time.time(); oc, orad = o.forward(); t_or = time.time() - t0
    sd = to_dev(sc)
    res = product_forward_raw(sd)
    torch.cuda.synchronize()
    print(f"P={P} {W}x{H} R oracle={o.num_rendered()} product={res['R']} oracle_fwd={t_or:.2f}s")
    print(" radii equal:", np.array_equal(res["radii"].cpu().numpy(), orad))
    for name in ["depths", "means2D", "conic_opacity", "rgb", "view2gaussian", "tiles_touched", "clamped"]:
        a = fetch(res, name); b = o.fetch(name)
        vis = orad > 0
        if name in ("depths",): a = a[vis]; b = b[vis]
        elif name in ("means2D", "conic_opacity", "rgb", "view2gaussian", "clamped"):
            per = a.size // P; a = a.reshape(P, per)[vis]; b = b.reshape(P, per)[vis]
        eq = np.array_equal(bits(a), bits(b))
        extra = "" if eq else f" maxrel={rel_err(a.astype(np.float64), b.astype(np.float64)).max():.3e} nmismatch={(bits(a)!=bits(b)).sum()}/{a.size}"
        print(f"  {name:14s} bit-exact={eq}{extra}")
    for name in ["point_list", "point_list_keys", "ranges", "n_contrib", "final_T"]:
        a = fetch(res, name); b = o.fetch(name)
        if a.dtype != b.dtype: a = a.view(b.dtype) if a.itemsize == b.itemsize else a.astype(b.dtype)
        eq = np.array_equal(bits(a), bits(b))
        extra = "" if eq else f" nmismatch={(bits(a)!=bits(b)).sum()}/{a.size}"
        print(f"  {name:14s} bit-exact={eq}{extra}")
    pc = res["color"].cpu().numpy()
    print("  out_color bit-exact:", np.array_equal(bits(pc), bits(oc)), "max abs diff", np.abs(pc - oc).max())
    # backward
    from diff_gaussian_rasterization import _backend as B
    rng = np.random.default_rng(1)
    dL = rng.normal(size=oc.shape).astype(np.float32)
    g_or = o.backward(dL)
    a = res["args"]
    grads = B.rasterize_gaussians_backward(a[0], a[1], res["radii"], a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                                           torch.from_numpy(dL).cuda(), a[17], a[18], a[19], res["geom"], res["R"], res["binning"], res["img"], False)
    torch.cuda.synchronize()
    names = ["means2D", "colors", "opacity", "means3D", "cov3D", "sh", "scales", "rotations", "view2gaussian"]
    for nme, g in zip(names, grads):
        gp = g.cpu().numpy().reshape(g_or[nme].shape); go = g_or[nme]
        denom = np.abs(go).max() + 1e-30
        print(f"  grad {nme:14s} max|diff|/max|ref|={np.abs(gp-go).max()/denom:.3e}  rel-l2={np.linalg.norm(gp-go)/(np.linalg.norm(go)+1e-30):.3e} nan={np.isnan(gp).sum()}")
    return sc, o, res

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run(2000, 128, 96, 100.0)
    run(20000, 320, 208, 240.0, ks=0.1)
    run(100000, 800, 528, 600.0, seed=3)
    # timing at S1M
    sc = S.scene_frustum(1_000_000, seed=0)
    sd = to_dev(sc)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        res = product_forward_raw(sd)
        torch.cuda.synchronize(); print("S1M forward ms", (time.time() - t0) * 1e3, "R", res["R"])
    from diff_gaussian_rasterization import _backend as B
    dL = torch.randn(9, sd["H"], sd["W"], device="cuda")
    a = res["args"]
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        grads = B.rasterize_gaussians_backward(a[0], a[1], res["radii"], a[2], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14],
                                               dL, a[17], a[18], a[19], res["geom"], res["R"], res["binning"], res["img"], False)
        torch.cuda.synchronize(); print("S1M backward ms", (time.time() - t0) * 1e3)
