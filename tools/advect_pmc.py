#!/usr/bin/env python3
"""Diagnostic: run only the advection kernels (cfg2 shape) so rocprofv3 --pmc can attribute cycles."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids
B, K, H, W = 32, 768, 32, 64
_, lg, og = make_grids(H, W, False)
geom = ops.AdvectGeometry(lg, og)
f = torch.randn(B, K, H, W, device="cuda", requires_grad=True)
vel = torch.randn(B, 2 * K, H, W, device="cuda", requires_grad=True)
go = torch.randn(B, K, H, W, device="cuda")
for _ in range(3):
    y = ops.sl_advect(f, vel[:, :K], vel[:, K:], geom, 0.196887, "bicubic")
    y.backward(go)
torch.cuda.synchronize()
