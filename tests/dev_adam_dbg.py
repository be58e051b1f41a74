import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_common import *
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, numpy as np
import train_epilogue as T
import train_epilogue_oracle as O
from test_train_epilogue_gpu import _make, _set_grads, GROUPS, DEV
pa, oa = _make(501, 4, T.FusedAdam); pb, ob = _make(501, 4, torch.optim.Adam)
po = [p.detach().cpu().numpy().copy() for p in pa]; mo = [np.zeros_like(x) for x in po]; vo = [np.zeros_like(x) for x in po]
def ostep(step, grads):
    for i in range(len(po)):
        po[i], mo[i], vo[i] = O.adam_step(po[i], grads[i], mo[i], vo[i], step, GROUPS[i][2])
def report(tag):
    print(tag, ["%.2e/%.2e" % (np.abs(a.detach().cpu().numpy() - o).max(), np.abs(b.detach().cpu().numpy() - o).max()) for a, b, o in zip(pa, pb, po)])
for step in range(2):
    _set_grads(pa, step, 5); _set_grads(pb, step, 5); oa.step(); ob.step()
    ostep(step + 1, [p.grad.cpu().numpy() for p in pa]); report("step%d" % step)
def surgery(opt):
    new = []
    for group in opt.param_groups:
        p = group["params"][0]
        ext = torch.full((10,) + tuple(p.shape[1:]), 0.25, device=DEV)
        st = opt.state.get(p, None)
        st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
        del opt.state[p]
        q = torch.cat((p.detach(), ext), dim=0)
        mask = torch.ones(q.shape[0], dtype=torch.bool, device=DEV); mask[::5] = False
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask], st["exp_avg_sq"][mask]
        q = torch.nn.Parameter(q[mask].requires_grad_(True))
        group["params"][0] = q
        opt.state[q] = st
        new.append(q)
    return new
pa, pb = surgery(oa), surgery(ob)
for i in range(len(po)):
    ext = np.full((10,) + po[i].shape[1:], 0.25, np.float32)
    mask = np.ones(po[i].shape[0] + 10, bool); mask[::5] = False
    po[i] = np.concatenate([po[i], ext])[mask]; mo[i] = np.concatenate([mo[i], 0 * ext])[mask]; vo[i] = np.concatenate([vo[i], 0 * ext])[mask]
report("surgery")
print("state m a/b", ["%.2e/%.2e" % (np.abs(oa.state[a]["exp_avg"].cpu().numpy() - m).max(), np.abs(ob.state[b]["exp_avg"].cpu().numpy() - m).max()) for a, b, m in zip(pa, pb, mo)])
for step in range(2, 4):
    _set_grads(pa, step, 5); _set_grads(pb, step, 5); oa.step(); ob.step()
    ostep(step + 1, [p.grad.cpu().numpy() for p in pa]); report("step%d" % step)
    print(" steps", [float(oa.state[p]["step"]) for p in pa], [float(ob.state[p]["step"]) for p in pb])
