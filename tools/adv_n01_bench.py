"""Diagnostic: the windowed advection at INCOHERENT large displacements (N(0, s) velocities over the full layer step,
128x256 B=8 K=768): forward / backward time per launch (round 3: 22.6 ms backward at s = 1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids
B, K, H, W = 8, 768, 128, 256
_, lg, og = make_grids(H, W, False)
geom = ops.AdvectGeometry(lg, og)
for scale in (0.02, 0.3, 1.0):
    f = torch.randn(B, K, H, W, device="cuda", requires_grad=True)
    vel = (torch.randn(B, 2 * K, H, W, device="cuda") * scale).requires_grad_(True)
    go = torch.randn(B, K, H, W, device="cuda")
    for name, flags in (("ring", None), ("strips-bwd", ops.advect_flags(strips=True)), ("tiles (r3)", ops.advect_flags(tiles=True))):
        ts = []
        for it in range(6):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            f.grad = None; vel.grad = None
            e[0].record()
            y = ops.sl_advect_vel(f, vel, geom, 0.196887, "bicubic", flags=flags)
            e[1].record()
            y.backward(go)
            e[2].record()
            torch.cuda.synchronize()
            ts.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
        ts = ts[2:]
        print("N(0,%.2f) %-11s fwd %.2f ms  bwd %.2f ms" % (scale, name, sum(t[0] for t in ts) / len(ts), sum(t[1] for t in ts) / len(ts)))
