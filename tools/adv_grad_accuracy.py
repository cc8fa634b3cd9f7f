"""Diagnostic: error of the advection operator's gradients against the fp64 oracle, next to the CPU-fp32 oracle's own
(rms and max, velocity gradients are ill-conditioned next to the poles).  PARADIS_HIP_LIB selects the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import paradis_oracle as O
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids


def rms(a, b):
    d = a.double() - b.double()
    return float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt())


def mx(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


name = os.path.basename(os.environ.get("PARADIS_HIP_LIB", "shipped")) + " flags=" + os.environ.get("PARADIS_ADVECT_FLAGS", "0")
for (H, W, poles, scale) in ((32, 64, False, 0.3), (32, 64, False, 0.05), (128, 256, False, 0.3), (128, 256, False, 0.05), (65, 130, True, 0.3)):
    B, K = 1, 6
    _, lg, og = make_grids(H, W, poles)
    g = torch.Generator().manual_seed(11)
    f, ct = torch.randn(B, K, H, W, generator=g), torch.randn(B, K, H, W, generator=g)
    u, v = torch.randn(B, K, H, W, generator=g) * scale, torch.randn(B, K, H, W, generator=g) * scale
    res = {}
    for dt in (torch.float64, torch.float32):
        fd, ud, vd = (t.detach().clone().to(dt).requires_grad_(True) for t in (f, u, v))
        y = O.sl_advect_core_aten(fd, ud, vd, 0.196887, O.GridGeometry(lg.to(dt), og.to(dt)), "bicubic")
        y.backward(ct.to(dt))
        res[dt] = (y.detach(), fd.grad, ud.grad, vd.grad)
    fc, uc, vc = (t.detach().clone().cuda().requires_grad_(True) for t in (f, u, v))
    y = ops.sl_advect(fc, uc, vc, ops.AdvectGeometry(lg, og), 0.196887, "bicubic")
    y.backward(ct.cuda())
    got = (y.detach().cpu(), fc.grad.cpu(), uc.grad.cpu(), vc.grad.cpu())
    r64, r32 = res[torch.float64], res[torch.float32]
    print("%-12s %dx%d scale %.2f | " % (name, H, W, scale) + " | ".join(
        "%s rms %.1e/%.1e max %.1e/%.1e" % (n, rms(a, c), rms(b, c), mx(a, c), mx(b, c))
        for n, a, b, c in zip(("y", "gf", "gu", "gv"), got, r32, r64)), flush=True)
