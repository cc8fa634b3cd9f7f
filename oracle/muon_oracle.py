"""CPU restatement of the Muon / NorMuon update the reference takes from the `dion` package
(reference trainer.py:337-364; requirements.txt:26 installs the un-pinned git HEAD of microsoft/dion).

TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED**: `dion` is not vendored in /root/reference and cannot
be installed here (no network), so no reference outputs exist to pin this file against; it restates
the algorithm as published in that repository (muon.py / normuon.py / newton_schulz_triton.py):
momentum, Frobenius normalisation, five quintic Newton-Schulz iterations with per-iteration
coefficients on the wide orientation, NorMuon's per-neuron second-moment normalisation with
Frobenius-norm preservation, shape-dependent learning-rate adjustment, decoupled weight decay.
`dion` evaluates the iteration in bfloat16; this oracle (and the HIP kernel) use float32.
"""
from __future__ import annotations

import math

import torch

NS_COEFFS = ((4.0848, -6.8946, 2.9270), (3.9505, -6.3029, 2.6377), (3.7418, -5.5913, 2.3037),
             (2.8769, -3.1427, 1.2046), (2.8366, -3.0525, 1.2012))


def newton_schulz(G: torch.Tensor, eps: float) -> torch.Tensor:
    X = G
    transposed = G.shape[-2] > G.shape[-1]
    if transposed:
        X = X.mT
    X = X / (X.norm() + eps)
    for a, b, c in NS_COEFFS:
        A = X @ X.mT
        B = b * A + c * (A @ A)
        X = a * X + B @ X
    return X.mT if transposed else X


def adjusted_lr(lr: float, shape, adjust: str | None) -> float:
    fan_out, fan_in = shape[0], math.prod(shape[1:])        # flatten=True (trainer.py:54)
    if adjust is None:
        return lr
    if adjust == "spectral_norm":
        return lr * math.sqrt(fan_out / fan_in)
    if adjust == "rms_norm":
        return lr * 0.2 * math.sqrt(max(fan_out, fan_in))
    raise ValueError(adjust)


def muon_step(W, G, M, lr, mu=0.95, weight_decay=0.01, eps=1e-8, nesterov=False, adjust="spectral_norm",
              V=None, beta2=0.95):
    """One update of a weight tensor (conv weights are flattened to [out, -1]).  M (and V for NorMuon,
    [out, 1]) are updated in place; returns the new weight.  V is not None selects NorMuon."""
    shape = W.shape
    w, g, m = W.reshape(shape[0], -1), G.reshape(shape[0], -1), M.view(shape[0], -1)
    m.mul_(mu).add_(g)
    u = g + mu * m if nesterov else m
    u = newton_schulz(u, eps)
    if V is not None:
        norm_u = u.norm()
        V.lerp_((u * u).mean(dim=-1, keepdim=True), 1 - beta2)
        u = u / (V.sqrt() + 1e-8)
        u = u * (norm_u / u.norm().clamp(min=1e-8))
    w = w * (1 - lr * weight_decay) - adjusted_lr(lr, shape, adjust) * u
    return w.reshape(shape)
