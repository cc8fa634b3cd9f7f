"""Diagnostic: 100 launches of the backward advection kernel at 32x64, B=32, K=768 after a clock warm-up
(for rocprofv3 --kernel-trace / --pmc).  argv[1] = velocity scale (default 0.05)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids
B, K, H, W = 32, 768, 32, 64
_, lg, og = make_grids(H, W, False)
geom = ops.AdvectGeometry(lg, og)
f = torch.randn(B, K, H, W, device="cuda")
vel = torch.randn(B, 2 * K, H, W, device="cuda") * float(sys.argv[1] if len(sys.argv) > 1 else 0.05)
go = torch.randn(B, K, H, W, device="cuda")
args = ops._geom_args(geom, f.device, 0.196887 / 8, "bicubic", None)
x = torch.randn(64 << 20, device="cuda")
for _ in range(3000):
    x = x * 1.0001
for _ in range(100):
    ops._sl_advect_vel_backward(go, f, vel, *args)
torch.cuda.synchronize()
