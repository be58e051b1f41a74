import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import synthetic_scenes as S, oracle_binding as ob
from test_parity_gpu import SCENES
for name in ["tiny", "small_ks01", "lego10k", "ragged", "long_lists"]:
    sc = SCENES[name]()
    o = ob.OracleScene(sc); oc, _ = o.forward()
    res = product_forward_raw(to_dev(sc)); pc = res["color"].cpu().numpy()
    d = np.abs(pc[8] - oc[8]); m = np.abs(oc[8]).max()
    rel = d / np.maximum(np.abs(oc[8]), 1e-30)
    fT = fetch(res, "final_T").reshape(4, -1); oT = o.fetch("final_T").reshape(4, -1)
    print(name, "dist max %.3e  abs diff max %.3e (%.2e of max)  rel>1e-4: %d px, worst rel %.2e at value %.3e" % (m, d.max(), d.max() / m, (rel > 1e-4).sum(), rel.max(), np.abs(oc[8]).ravel()[rel.argmax()]),
          " raw-dist diff/max %.2e  dist1 %.2e dist2 %.2e normals %.2e" % (np.abs(fT[3] - oT[3]).max() / max(np.abs(oT[3]).max(), 1e-30), np.abs(fT[1] - oT[1]).max() / np.abs(oT[1]).max(), np.abs(fT[2] - oT[2]).max() / np.abs(oT[2]).max(), np.abs(pc[3:6] - oc[3:6]).max()))
