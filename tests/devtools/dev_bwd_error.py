"""Developer script: actual gradient error of the product's backward vs the oracle on a few scenes (tests assert <= 1e-4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpu_common import *
import test_parity_gpu as TP
for name in ("lego10k", "ragged", "stress_box", "mid100k"):
    sc = TP.SCENES[name]()
    o, oc, orad, res = TP._forward_pair(sc)
    dL = np.random.default_rng(1).normal(size=oc.shape).astype(np.float32)
    go = o.backward(dL); gp = TP._product_backward(res, dL)
    print(name, {k: "%.2e" % (np.abs(gp[k].reshape(go[k].shape) - go[k]).max() / np.abs(go[k]).max()) for k in ("means2D", "colors", "opacity", "view2gaussian")})
