"""Accuracy of training_loss's gradient vs the reference composition in fp32 (oracle, CPU), both against float64."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("oracle", "gaussian-opacity-fields_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import train_epilogue_oracle as O
import test_train_epilogue_gpu as TT
for (W, H, seed) in ((1600, 1063, 5863), (1600, 1063, 1), (800, 531, 2)):
    r, gt, wvt = TT._loss_case(W, H, seed)
    lambdas = (0.2, 0.05, 100.0)
    terms, gp = TT._training_loss_product(r.numpy(), gt.numpy(), wvt.numpy(), W, H, 0.85, 0.6, lambdas)
    ro = r.clone().requires_grad_(True)
    O.training_loss(ro, gt, wvt, W, H, 0.85, 0.6, *lambdas)[0].backward()
    go = ro.grad.numpy()
    g64 = TT._loss_grad64(r, gt, wvt, W, H, 0.85, 0.6, lambdas)
    cl = np.sqrt((r[3:6].double().numpy() ** 2).sum(0)) < 1e-12
    for c in (3, 4, 5, 6):
        m = ~cl if c != 6 else np.ones_like(cl)
        ep, er = (gp[c] - g64[c])[m], (go[c] - g64[c])[m]
        print(W, H, seed, "ch", c, "max ours %.3e ref %.3e | rms ours %.3e ref %.3e | scale %.3e" % (np.abs(ep).max(), np.abs(er).max(), np.sqrt((ep**2).mean()), np.sqrt((er**2).mean()), np.abs(g64[c][m]).max()))
