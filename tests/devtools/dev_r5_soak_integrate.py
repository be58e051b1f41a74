"""Round 5 soak of the opacity-field query's two pixel-pass forms on the GPU: ray-centric (shipped; waves leave the kernel early) vs
pixel-centric, every output bit compared, on N random small scenes (the fuzz recipe of tests/test_parity_gpu.py, seeds 1000+) and on the
full-size scenes S1M / S1M posed / S1M-clustered.  A kernel that hangs shows up as this script's timeout, not as a hung box:
    timeout 400 python tests/devtools/dev_r5_soak_integrate.py 400"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import to_dev, settings_from, bits
import synthetic_scenes as S
import test_parity_gpu as TP
from diff_gaussian_rasterization import GaussianRasterizer, _backend as B

def both(sc, pts):
    sd = to_dev(sc)
    p = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)).cuda()
    outs = []
    for mode in (0, 1):
        prev = B.lib.gof_set_integrate_pixel_pass(mode)
        try:
            r = GaussianRasterizer(settings_from(sd))
            o = r.integrate(points3D=p, means3D=sd["means3D"], means2D=None, opacities=sd["opacities"], shs=sd["shs"], scales=sd["scales"], rotations=sd["rotations"])
            torch.cuda.synchronize()
        finally:
            B.lib.gof_set_integrate_pixel_pass(prev)
        outs.append([t.cpu().numpy() for t in o])
    return all(np.array_equal(bits(a), bits(b)) for a, b in zip(*outs))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.time(); bad = []
for seed in range(1000, 1000 + n):
    sc = TP._fuzz_scene(seed)
    if not both(sc, S.tetra_points(sc)[:20000]):
        bad.append(seed)
print("fuzz scenes: %d, differing: %s, %.1f s" % (n, bad, time.time() - t0), flush=True)
for name, mk in (("s1m", lambda: S.scene_frustum(1_000_000, seed=0)), ("s1m_posed", lambda: S.scene_frustum(1_000_000, seed=0, pose_seed=0)),
                 ("s1m_clustered", lambda: S.scene_clustered(1_000_000, seed=0)), ("s1m_ks01_small_image", lambda: S.scene_frustum(1_000_000, W=801, H=533, focal=600.0, seed=3, kernel_size=0.1))):
    sc = mk()
    ok = both(sc, S.tetra_points(sc)[::9])
    print(name, "identical" if ok else "DIFFERENT", flush=True)
    if not ok: bad.append(name)
sys.exit(1 if bad else 0)
