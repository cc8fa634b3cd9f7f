"""AdamW step on the HIP device (reference ``trainer.py:327-335`` uses
``torch.optim.AdamW(params, lr, weight_decay, betas)``).  Same update rule and operation order as
``torch.optim.AdamW`` (decoupled weight decay, bias-corrected moments, eps added after the sqrt);
state-dict layout compatible with it (``step``, ``exp_avg``, ``exp_avg_sq``)."""
import torch

from ._lib import check, dptr, lib, require_hip, stream_ptr


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = stream_ptr()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                require_hip(p, p.grad)
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                check(lib.paradis_adamw_step(dptr(p), dptr(g), dptr(state["exp_avg"]),
                                             dptr(state["exp_avg_sq"]), p.numel(), group["lr"], b1, b2,
                                             group["eps"], group["weight_decay"], int(state["step"]), st),
                      "adamw_step")
        return loss
