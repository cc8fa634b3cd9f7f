#!/usr/bin/env python3
"""Diagnostic: run only the advection kernels so rocprofv3 --pmc can attribute cycles.
  python tools/advect_pmc.py [HxW [B [K [velocity sigma]]]]     (default: the cfg2 shape 32x64, B = 32, K = 768, sigma 1)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids
a = sys.argv[1:]
H, W = (int(v) for v in (a[0] if a else "32x64").split("x"))
B = int(a[1]) if len(a) > 1 else 32
K = int(a[2]) if len(a) > 2 else 768
sigma = float(a[3]) if len(a) > 3 else 1.0
_, lg, og = make_grids(H, W, H % 2 == 1)
geom = ops.AdvectGeometry(lg, og)
f = torch.randn(B, K, H, W, device="cuda", requires_grad=True)
vel = (torch.randn(B, 2 * K, H, W, device="cuda") * sigma).requires_grad_(True)
go = torch.randn(B, K, H, W, device="cuda")
for _ in range(3):
    y = ops.sl_advect(f, vel[:, :K], vel[:, K:], geom, 0.196887, "bicubic")
    y.backward(go)
torch.cuda.synchronize()
