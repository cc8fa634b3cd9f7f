#!/usr/bin/env python3
"""Bitwise comparison of the bf16x3 pointwise GEMM (forward, dX, dW) over the layer shapes between two library builds or
settings: run once per setting (`PARADIS_HIP_LIB=... python tools/quad_check.py out.pt [reference.pt]`).  Round 6 used it
for the 256 x 256 'quad' tile (profiles/r06_gemm_quad.txt)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops
out = sys.argv[1]
g = torch.Generator().manual_seed(3)
res = {}
for (B, Ci, Co, H, W) in ((2, 1024, 1024, 32, 64), (2, 1152, 896, 32, 64), (1, 896, 896, 32, 64), (2, 384, 1536, 32, 64),
                          (2, 1024, 384, 32, 64), (1, 186, 1024, 32, 64), (1, 1024, 97, 32, 64), (1, 768, 1024, 128, 256),
                          (3, 130, 258, 9, 20)):
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda()
    ct = torch.randn(B, Co, H, W, generator=g).cuda()
    y = ops.pointwise(x, w, b, act="SiLU", scheme=ops.GEMM_BF16X3)
    y.backward(ct)
    res[(B, Ci, Co, H, W)] = (y.detach().cpu(), x.grad.cpu(), w.grad.cpu())
torch.save(res, out)
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k in res:
        eq = [bool(torch.equal(a, b)) for a, b in zip(res[k], ref[k])]
        print(k, "bitwise equal (y, gx, gw):", eq)
