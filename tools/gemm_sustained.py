#!/usr/bin/env python3
"""Diagnostic: is the in-step GEMM rate (120 TF) a clock/thermal effect or a cache effect?
(a) one 1024x1024x65536 forward GEMM back to back for ~1.5 s, rate per 100-launch window;
(b) the same GEMM alternating with a streaming kernel over 1 GB (evicts L2 / Infinity Cache)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd._lib import dptr, lib, stream_ptr

B, P, Co, Ci = 32, 2048, 1024, 1024
w = torch.randn(Co, Ci, device="cuda") * Ci ** -0.5
wt = w.t().contiguous()
x = torch.randn(B, Ci, P, device="cuda"); y = torch.empty(B, Co, P, device="cuda")
big = torch.randn(256 << 20, device="cuda"); big2 = torch.empty_like(big)
st = stream_ptr()
flops = 2.0 * B * Co * Ci * P
gemm = lambda: lib.paradis_pw_gemm_fwd(dptr(w), dptr(wt), None, 0, None, dptr(x), None, None, None, None, 0, None, dptr(y), None, B, Co, Ci, P, Ci * P, 0, Co * P, 0, st)
for _ in range(5):
    gemm()
torch.cuda.synchronize()
print("(a) back to back:")
for win in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        gemm()
    e1.record(); torch.cuda.synchronize()
    print(f"   window {win:2d}: {flops * 100 / e0.elapsed_time(e1) / 1e9:6.1f} TF")
print("(b) alternating with a 2 GB stream copy (GEMM timed alone with events):")
tot = 0.0
evs = []
for i in range(60):
    big2.copy_(big)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gemm(); e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
ts = [a.elapsed_time(b) for a, b in evs][10:]
print(f"   mean {flops / (sum(ts) / len(ts)) / 1e9:6.1f} TF, best {flops / min(ts) / 1e9:6.1f}, worst {flops / max(ts) / 1e9:6.1f}")
