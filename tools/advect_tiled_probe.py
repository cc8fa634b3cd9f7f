#!/usr/bin/env python3
"""Diagnostic: a few tiled advection calls (128x256, small displacements) for rocprofv3 --kernel-trace."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops
from paradis_model_amd._lib import lib
from paradis_model_amd.harness import make_grids
B, K, H, W = 8, 768, 128, 256
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
_, lg, og = make_grids(H, W, False)
geom = ops.AdvectGeometry(lg, og)
f = torch.randn(B, K, H, W, device="cuda", requires_grad=True)
vel = (torch.randn(B, 2 * K, H, W, device="cuda") * scale).requires_grad_(True)
go = torch.randn(B, K, H, W, device="cuda")
for _ in range(4):
    y = ops.sl_advect(f, vel[:, :K], vel[:, K:], geom, 0.196887, "bicubic")
    y.backward(go)
torch.cuda.synchronize()
