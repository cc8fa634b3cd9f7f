"""Diagnostic: which outputs of the windowed advection backward differ between two identical calls (PARADIS_DETERMINISTIC=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids
H, W, B, K = 128, 256, 1, 3
_, lg, og = make_grids(H, W, False)
g = torch.Generator().manual_seed(3)
f, ct = torch.randn(B, K, H, W, generator=g), torch.randn(B, K, H, W, generator=g)
u, v = torch.randn(B, K, H, W, generator=g) * 0.5, torch.randn(B, K, H, W, generator=g) * 0.5
geo = ops.AdvectGeometry(lg, og)
for flags in (None, ops.advect_flags(strips=True), ops.advect_flags(tiles=True)):
    outs = []
    for _ in range(3):
        fc, uc, vc = (t.cuda().requires_grad_(True) for t in (f, u, v))
        y = ops.sl_advect(fc, uc, vc, geo, 0.196887, "bicubic", flags=flags)
        y.backward(ct.cuda())
        outs.append((y.detach().clone(), fc.grad.clone(), uc.grad.clone(), vc.grad.clone()))
    for name, i in (("out", 0), ("gf", 1), ("gu", 2), ("gv", 3)):
        for k in (1, 2):
            d = (outs[0][i] != outs[k][i])
            if bool(d.any()):
                idx = d.nonzero()[:5].tolist()
                print(flags, name, "run0 vs run%d: %d differ, max |d| %.3e" % (k, int(d.sum()), float((outs[0][i] - outs[k][i]).abs().max())), idx)
    print(flags, "checked")
