"""FusedAdam: the optimizer ``scene/gaussian_model.py:360`` constructs (``torch.optim.Adam(l, lr=0.0, eps=1e-15)``,
one param group per Gaussian attribute with its own lr) with the whole step in ONE HIP launch over all groups
(28 B of HBM traffic per parameter float) instead of torch's ~12 foreach launches.

State layout is torch's ("step" a CPU fp32 scalar tensor, "exp_avg", "exp_avg_sq"), so the reference's densification
code that edits ``optimizer.state`` in place (gaussian_model.py:532-607) and its checkpoints (``state_dict`` /
``load_state_dict``, :130,150) work unchanged.  Arithmetic follows torch/optim/adam.py ``_multi_tensor_adam``."""
import torch

from . import _backend as B


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
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                step = float(state["step"])
                for k in ("exp_avg", "exp_avg_sq"):
                    if not state[k].is_contiguous():
                        state[k] = state[k].contiguous()
                bias_correction1 = 1 - beta1 ** step
                bias_correction2 = 1 - beta2 ** step
                step_size = -(lr / bias_correction1)
                launches.setdefault((float(beta1), float(beta2), eps), []).append(
                    (p, p.grad.contiguous(), state["exp_avg"], state["exp_avg_sq"], step_size, bias_correction2 ** 0.5))
        for (beta1, beta2, eps), entries in launches.items():
            B.adam_step(entries, beta1, beta2, eps)
        return loss
