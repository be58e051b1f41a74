"""FusedAdam: the optimizer ``scene/gaussian_model.py:360`` constructs (``torch.optim.Adam(l, lr=0.0, eps=1e-15)``,
one param group per Gaussian attribute with its own lr) with the whole step in ONE HIP launch over all groups
(28 B of HBM traffic per parameter float) instead of torch's ~12 foreach launches.

State layout is torch's ("step" a CPU fp32 scalar tensor, "exp_avg", "exp_avg_sq"), so the reference's densification
code that edits ``optimizer.state`` in place (gaussian_model.py:532-607) and its checkpoints (``state_dict`` /
``load_state_dict``, :130,150) work unchanged.  Arithmetic follows torch/optim/adam.py ``_multi_tensor_adam``."""
import torch

from . import _backend as B


def _is_dense(t):
    """Non-overlapping and dense: the strides are a permutation of a contiguous layout of the same sizes."""
    expect = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz != 1), key=lambda x: x[1]):
        if stride != expect:
            return False
        expect *= size
    return True


def _flat(t):
    """The tensor's elements in MEMORY order as a 1-D view (all operands of one update share the layout, so the order is irrelevant)."""
    return t.view(-1) if t.is_contiguous() else t.as_strided((t.numel(),), (1,), t.storage_offset())


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None):
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError("FusedAdam: tensor lr is not supported")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        if weight_decay != 0 or amsgrad or maximize or capturable or differentiable:
            raise NotImplementedError("FusedAdam implements the configuration the reference uses: no weight decay, no amsgrad, "
                                      "no maximize/capturable/differentiable")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                        foreach=foreach, capturable=capturable, differentiable=differentiable, fused=fused)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        launches = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not implemented")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam: parameters must be float32 on a ROCm device (got %s, %s)" % (p.dtype, p.device))
                # Adam is element-wise, so any dense layout works as long as parameter, gradient and moments share it.  The reference
                # does hand over a non-contiguous parameter: _xyz is built from `np.vstack([x, y, z]).T` (dataset_readers.py:115,
                # gaussian_model.py:317, 339), i.e. column-major, until the first densification re-creates it.
                dense = p.is_contiguous() or _is_dense(p)
                if not dense:
                    raise RuntimeError("FusedAdam: parameters must be dense (non-overlapping, no gaps); got strides %s for shape %s"
                                       % (tuple(p.stride()), tuple(p.shape)))
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                step = float(state["step"])
                for k in ("exp_avg", "exp_avg_sq"):               # e.g. moments restored from a checkpoint with another layout
                    if state[k].stride() != p.stride() or state[k].shape != p.shape:
                        state[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(state[k])
                grad = p.grad
                if grad.stride() != p.stride():
                    grad = torch.empty_like(p, memory_format=torch.preserve_format).copy_(grad)
                bias_correction1 = 1 - beta1 ** step
                bias_correction2 = 1 - beta2 ** step
                step_size = -(lr / bias_correction1)
                launches.setdefault((float(beta1), float(beta2), eps), []).append(
                    (_flat(p), _flat(grad), _flat(state["exp_avg"]), _flat(state["exp_avg_sq"]), step_size, bias_correction2 ** 0.5))
        for (beta1, beta2, eps), entries in launches.items():
            B.adam_step(entries, beta1, beta2, eps)
        return loss
