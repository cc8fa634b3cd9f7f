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
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            steps = set()
            for p in params:
                require_hip(p, p.grad)
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                steps.add(int(state["step"]))
            uniform = len(steps) == 1 and all(p.is_contiguous() and p.grad.is_contiguous() for p in params)
            if uniform and len(params) > 1:
                self._step_group_fused(gi, group, params, steps.pop(), st)
                continue
            for p in params:
                state = self.state[p]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                check(lib.paradis_adamw_step(dptr(p), dptr(g), dptr(state["exp_avg"]),
                                             dptr(state["exp_avg_sq"]), p.numel(), group["lr"], b1, b2,
                                             group["eps"], group["weight_decay"], int(state["step"]), st),
                      "adamw_step")
        return loss

    def _step_group_fused(self, gi, group, params, step, st):
        """One launch for the group (``paradis_adamw_multi``).  The chunk list and the parameter /
        moment addresses are built once per parameter set; the gradient addresses are refreshed every
        step (autograd re-allocates ``.grad`` unless DDP keeps bucket views) through a pinned staging
        buffer, without a host synchronisation."""
        dev = params[0].device
        key = tuple(id(p) for p in params)
        cache = self.__dict__.setdefault("_fused_cache", {})
        c = cache.get(gi)
        if c is None or c["key"] != key:
            T = len(params)
            chunk = lib.paradis_adamw_chunk()
            ct, co = [], []
            for t, p in enumerate(params):
                for off in range(0, p.numel(), chunk):
                    ct.append(t)
                    co.append(off)
            host = torch.empty(4 * T, dtype=torch.int64).pin_memory()
            for t, p in enumerate(params):
                host[t] = p.data_ptr()
                host[2 * T + t] = self.state[p]["exp_avg"].data_ptr()
                host[3 * T + t] = self.state[p]["exp_avg_sq"].data_ptr()
            c = cache[gi] = dict(
                key=key, T=T, host=host, ptrs=torch.empty(4 * T, dtype=torch.int64, device=dev),
                numel=torch.tensor([p.numel() for p in params], dtype=torch.int64, device=dev),
                chunk_tensor=torch.tensor(ct, dtype=torch.int32, device=dev),
                chunk_off=torch.tensor(co, dtype=torch.int64, device=dev), n_chunks=len(ct))
        T, host = c["T"], c["host"]
        if c.get("pending") is not None:     # the previous step's async copy out of `host` (long done)
            c["pending"].synchronize()
        host[T:2 * T] = torch.tensor([p.grad.data_ptr() for p in params], dtype=torch.int64)
        c["ptrs"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        c["pending"] = ev
        b1, b2 = group["betas"]
        check(lib.paradis_adamw_multi(dptr(c["ptrs"]), dptr(c["numel"]), dptr(c["chunk_tensor"]),
                                      dptr(c["chunk_off"]), T, c["n_chunks"], group["lr"], b1, b2, group["eps"],
                                      group["weight_decay"], step, st), "adamw_multi")
