"""AdamW step on the HIP device (reference ``trainer.py:327-335`` uses
``torch.optim.AdamW(params, lr, weight_decay, betas)``).  Same update rule and operation order as
``torch.optim.AdamW`` (decoupled weight decay, bias-corrected moments, eps added after the sqrt);
state-dict layout compatible with it (``step``, ``exp_avg``, ``exp_avg_sq``)."""
import ctypes

import torch

from . import ops
from ._lib import check, dptr, lib, require_hip, stream_ptr


class AdamW(torch.optim.Optimizer):
    """``capturable=True``: the step count and the learning rate of every group also live on the device
    (``paradis_adamw_multi(..., dev_state)``), so a HIP graph captured around ``step()`` stays valid from replay to
    replay (``harness.GraphedTrainStep``); same formula, the bias corrections formed in double on the device (agrees with the
    host-side path to ~1 ulp of the fp32 corrections: the device pow is not the host libm, test bound 1e-6)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = bool(capturable)
        self._dev_state = {}      # group index -> (int32[2] device tensor, lr it holds)

    def _device_state(self, gi, group, dev, step_before):
        """int32[2] = [step count, bits of lr] of group ``gi`` on the device (created at the group's first update)"""
        import struct
        ent = self._dev_state.get(gi)
        lr_bits = struct.unpack("<i", struct.pack("<f", float(group["lr"])))[0]
        if ent is None:
            ent = [torch.tensor([step_before, lr_bits], dtype=torch.int32, device=dev), float(group["lr"])]
            self._dev_state[gi] = ent
        elif ent[1] != float(group["lr"]):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("AdamW(capturable): the learning rate changed inside a graph capture")
            ent[0][1:2].copy_(torch.tensor([lr_bits], dtype=torch.int32), non_blocking=False)
            ent[1] = float(group["lr"])
        return ent[0]

    def sync_device_state(self):
        """push the host-side learning rates to the device (call between graph replays after a scheduler step)"""
        for gi, group in enumerate(self.param_groups):
            if gi in self._dev_state:
                self._device_state(gi, group, self._dev_state[gi][0].device, 0)

    def note_replayed(self):
        """a captured step() was replayed: advance the host-side step counts (state_dict compatibility)"""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] += 1

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
            if uniform and (len(params) > 1 or self.capturable):
                self._step_group_fused(gi, group, params, steps.pop(), st)
                continue
            if self.capturable:
                raise RuntimeError("AdamW(capturable) needs contiguous parameters with one common step count per group")
            for p in params:
                state = self.state[p]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                check(lib.paradis_adamw_step(dptr(p), dptr(g), dptr(state["exp_avg"]),
                                             dptr(state["exp_avg_sq"]), p.numel(), group["lr"], b1, b2,
                                             group["eps"], group["weight_decay"], int(state["step"]), st),
                      "adamw_step")
        ops.weights_updated()     # the kernels write through raw pointers: no version-counter bump
        return loss

    def snapshot_pointer_tables(self):
        """copies of the pinned address tables of the fused groups (``harness.GraphedTrainStep``: a captured step
        re-reads them on every replay)"""
        return {gi: c["host"].clone() for gi, c in self.__dict__.get("_fused_cache", {}).items()}

    def restore_pointer_tables(self, tables) -> None:
        for gi, t in tables.items():
            c = self.__dict__.get("_fused_cache", {}).get(gi)
            if c is not None and c["host"].numel() == t.numel():
                if c.get("pending") is not None:
                    c["pending"].synchronize()
                c["host"].copy_(t)

    def _step_group_fused(self, gi, group, params, step, st):
        """One launch for the group (``paradis_adamw_multi``).  Only the chunk list (a function of the
        parameter sizes) is cached; the four address rows (parameter, gradient, both moments) are
        rewritten every step - ``load_state_dict``, ``p.data = ...`` or ``model.to()`` replace tensors
        behind the same parameter ids - and reach the device through a pinned staging buffer without a
        host synchronisation."""
        dev = params[0].device
        key = tuple((id(p), p.numel()) for p in params) + (str(dev),)
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
            c = cache[gi] = dict(
                key=key, T=T, host=host, ptrs=torch.empty(4 * T, dtype=torch.int64, device=dev),
                numel=torch.tensor([p.numel() for p in params], dtype=torch.int64, device=dev),
                chunk_tensor=torch.tensor(ct, dtype=torch.int32, device=dev),
                chunk_off=torch.tensor(co, dtype=torch.int64, device=dev), n_chunks=len(ct))
        T, host = c["T"], c["host"]
        capturing = torch.cuda.is_current_stream_capturing()
        if c.get("pending") is not None and not capturing:     # the previous step's async copy out of `host` (long done)
            c["pending"].synchronize()
        moments = [(self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in params]
        for m, v in moments:
            require_hip(m, v)
            if not (m.is_contiguous() and v.is_contiguous()):
                raise RuntimeError("AdamW: non-contiguous optimizer state")
        host.copy_(torch.tensor([p.data_ptr() for p in params] + [p.grad.data_ptr() for p in params]
                                + [m.data_ptr() for m, _ in moments] + [v.data_ptr() for _, v in moments],
                                dtype=torch.int64))
        c["ptrs"].copy_(host, non_blocking=True)
        if capturing:
            c["pending"] = None      # (inside a capture the copy is a graph node; nothing to wait for on the host)
        else:
            ev = torch.cuda.Event()
            ev.record()
            c["pending"] = ev
        b1, b2 = group["betas"]
        dev_state = None
        if self.capturable:
            dev_state = self._device_state(gi, group, dev, step - 1)
            check(lib.paradis_adamw_tick(dptr(dev_state), st), "adamw_tick")
        check(lib.paradis_adamw_multi(dptr(c["ptrs"]), dptr(c["numel"]), dptr(c["chunk_tensor"]),
                                      dptr(c["chunk_off"]), T, c["n_chunks"], group["lr"], b1, b2, group["eps"],
                                      group["weight_decay"], step, dptr(dev_state), st), "adamw_multi")


# ---------------------------------------------------------------------------------------------
# Muon / NorMuon (the reference's default optimiser family, trainer.py:337-364, from `dion`)
# ---------------------------------------------------------------------------------------------
def build_param_groups(model, lr, weight_decay, optimizer_name):
    """The reference's split of the parameters between the two algorithms (rule of reference
    ``trainer.py:24-64``): a ``weight`` owned directly by a Linear / ConvNd module is a matrix for the
    Muon family (conv kernels flattened to 2-D); every other trainable parameter - the biases of those
    modules first, then norm scales, GlobalBias factors, gates - is updated by AdamW.  Group order and
    the order inside each group follow module traversal, so optimiser state dicts index parameters
    the way the reference's do."""
    from torch import nn
    matrix_owner = (nn.Linear, nn.Conv1d, nn.Conv2d, nn.Conv3d)
    role = {}                                  # id(parameter) -> "matrix" | "bias": first owner wins
    for module in model.modules():
        if isinstance(module, matrix_owner):
            for name, kind in (("weight", "matrix"), ("bias", "bias")):
                p = module._parameters.get(name)
                if p is not None:
                    role.setdefault(id(p), kind)
    buckets = {"matrix": [], "bias": [], None: []}
    for p in model.parameters():               # de-duplicated, module traversal order
        if p.requires_grad:
            buckets[role.get(id(p))].append(p)
    return [dict(params=buckets["matrix"], algorithm=optimizer_name, lr=lr, weight_decay=weight_decay,
                 flatten=True),
            dict(params=buckets["bias"] + buckets[None], algorithm="adamw", lr=lr, weight_decay=weight_decay)]


def _adjusted_lr(lr, shape, adjust):
    import math
    fan_out, fan_in = shape[0], math.prod(shape[1:])
    if adjust is None:
        return lr
    if adjust == "spectral_norm":
        return lr * math.sqrt(fan_out / fan_in)
    if adjust == "rms_norm":
        return lr * 0.2 * math.sqrt(max(fan_out, fan_in))
    raise ValueError(f"unknown adjust_lr {adjust!r}")


class Muon(AdamW):
    """Muon with the constructor of ``dion.Muon`` as the reference calls it (trainer.py:347-354):
    parameter groups carry ``algorithm`` ("muon" / "normuon" for 2-D-flattened weights, "adamw" for
    the rest, see ``build_param_groups``).  The matrix update runs in one C-ABI call per weight
    (``paradis_muon_step``: fp32 Newton-Schulz on the GEMMs of ``ops`` - bf16x3 split unless ``ops.GEMM_SCHEME`` is exact - no Triton); AdamW groups use the fused kernel
    of the base class.  ``dion`` is neither vendored nor pinned by the reference: the algorithm is
    restated from its published form (oracle/muon_oracle.py), parity unpinned."""

    _NORMUON = False
    _DEFAULT_ADJUST = "spectral_norm"

    def __init__(self, params, lr=0.01, mu=0.95, betas=(0.9, 0.95), weight_decay=0.01, epsilon=1e-8,
                 nesterov=False, adjust_lr="default", flatten=False, use_triton=False, muon_beta2=0.95):
        del use_triton   # accepted for signature compatibility; there is no Triton on this path
        if adjust_lr == "default":
            adjust_lr = self._DEFAULT_ADJUST
        params = list(params)
        if params and not isinstance(params[0], dict):
            params = [dict(params=params, algorithm="normuon" if self._NORMUON else "muon")]
        for gdict in params:
            gdict.setdefault("algorithm", "normuon" if self._NORMUON else "muon")
        super().__init__(params, lr=lr, betas=betas, eps=epsilon, weight_decay=weight_decay)
        for group in self.param_groups:
            group.setdefault("mu", mu)
            group.setdefault("nesterov", nesterov)
            group.setdefault("adjust_lr", adjust_lr)
            group.setdefault("flatten", flatten)
            group.setdefault("muon_beta2", muon_beta2)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        adamw_groups = [g for g in self.param_groups if g["algorithm"] == "adamw"]
        matrix_groups = [g for g in self.param_groups if g["algorithm"] != "adamw"]
        for gi, group in enumerate(matrix_groups):
            if group["algorithm"] not in ("muon", "normuon"):
                raise ValueError(f"unknown algorithm {group['algorithm']!r}")
            self._step_matrix_group(gi, group)
        if adamw_groups:
            saved = self.param_groups
            self.param_groups = adamw_groups
            try:
                AdamW.step(self)
            finally:
                self.param_groups = saved
        return loss

    def _step_matrix_group(self, gi, group):
        """Same-shaped matrices are updated together (one ``paradis_muon_step`` per shape: the batched
        Newton-Schulz GEMMs fill the chip).  One device table of the w / g / momentum / variance
        addresses serves all shapes; the gradient addresses are refreshed each step through a pinned
        staging buffer (no host synchronisation)."""
        normuon = group["algorithm"] == "normuon"
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return
        for p in params:
            require_hip(p, p.grad)
            if p.dim() < 2:
                raise ValueError("Muon parameters must be matrices (use an adamw group for the rest)")
            if p.dim() > 2 and not group["flatten"]:
                raise ValueError("conv weights need flatten=True (reference trainer.py:54)")
            if not p.is_contiguous():
                raise ValueError("Muon: non-contiguous parameter")
            state = self.state[p]
            if not state:
                state["step"] = 0
                state["momentum"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if normuon:
                    state["variance_neuron"] = torch.zeros(p.shape[0], 1, dtype=p.dtype, device=p.device)
            state["step"] += 1
        dev = params[0].device
        key = tuple((id(p), tuple(p.shape)) for p in params) + (str(dev), normuon)
        cache = self.__dict__.setdefault("_muon_cache", {})
        c = cache.get(gi)
        if c is None or c["key"] != key:
            by_shape = {}
            for p in params:
                by_shape.setdefault((p.shape[0], p.numel() // p.shape[0], tuple(p.shape)), []).append(p)
            order, shapes = [], []
            for (rows, cols, full), ps in by_shape.items():
                shapes.append((rows, cols, full, len(order), len(ps)))
                order.extend(ps)
            T = len(order)
            host = torch.zeros(4 * T, dtype=torch.int64).pin_memory()
            ws_bytes = max(lib.paradis_muon_ws_bytes(n, rows, cols) for rows, cols, _, _, n in shapes)
            c = cache[gi] = dict(key=key, order=order, shapes=shapes, T=T, host=host,
                                 table=torch.empty(4 * T, dtype=torch.int64, device=dev),
                                 ws=torch.empty(ws_bytes // 4 + 64, dtype=torch.float32, device=dev))
        T, host = c["T"], c["host"]
        if c.get("pending") is not None:
            c["pending"].synchronize()
        grads = []
        for p in c["order"]:
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            grads.append(g)          # keep alive until the kernels are queued
        # all four address rows are rewritten every step (state tensors can be replaced by
        # load_state_dict behind the same parameter ids)
        mom = [self.state[p]["momentum"] for p in c["order"]]
        var = [self.state[p]["variance_neuron"] for p in c["order"]] if normuon else []
        require_hip(*mom, *var)
        host.copy_(torch.tensor([p.data_ptr() for p in c["order"]] + [g.data_ptr() for g in grads]
                                + [m.data_ptr() for m in mom]
                                + ([v.data_ptr() for v in var] if normuon else [0] * T), dtype=torch.int64))
        c["table"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        c["pending"] = ev
        st = stream_ptr()
        lr = group["lr"]
        base = c["table"].data_ptr()
        for rows, cols, full, off, n in c["shapes"]:
            check(lib.paradis_muon_step(ctypes.c_void_p(base + 8 * off), T, n, rows, cols, lr,
                                        _adjusted_lr(lr, full, group["adjust_lr"]), group["mu"],
                                        group["muon_beta2"], group["weight_decay"], group["eps"],
                                        int(bool(group["nesterov"])), int(normuon),
                                        1 if ops.GEMM_SCHEME != ops.GEMM_EXACT else 0, dptr(c["ws"]), st),
                  "muon_step")
        ops.weights_updated()


class NorMuon(Muon):
    """``dion.NorMuon`` as called at reference trainer.py:355-362 (the shipped default,
    config/paradis_settings.yaml:117): Muon + per-neuron second-moment normalisation."""

    _NORMUON = True
    _DEFAULT_ADJUST = "rms_norm"
