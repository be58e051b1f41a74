import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S
from diff_gaussian_rasterization import _backend as B
sc = S.scene_frustum(1_000_000, seed=0)
sd = to_dev(sc)
out = (C.c_ulonglong * 8)()
B.lib.gof_debug_fw_stats(out, 1)
res = product_forward_raw(sd); torch.cuda.synchronize()
B.lib.gof_debug_fw_stats(out, 1)
s = list(out)
npix = sd["W"] * sd["H"]; nwaves = 6700 * 4
print("scanned wave-entries %d (per wave %.1f)" % (s[0], s[0] / nwaves))
print("candidates %d (per pixel %.1f), exact-pass %d (per pixel %.1f), contributing %d (per pixel %.1f)" % (s[1], s[1] / npix, s[3], s[3] / npix, s[4], s[4] / npix))
print("(cull audit build only) pairs accepted by the exact path that the scan had dropped (must be 0):", s[6])
print("phase-2 wave iterations %d (per wave %.1f); lane-iterations active %d -> lane utilisation %.2f" % (s[2], s[2] / nwaves, s[5], s[5] / (64.0 * s[2])))
