#!/usr/bin/env python3
"""Diagnostic: error of the HIP advection (forward + gradients) against fp64 autograd through the
oracle, next to the CPU-fp32 oracle's own error (the SURVEY 8c-iii yardstick).
Usage on the GPU box: python tools/advect_accuracy.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import paradis_oracle as O  # noqa: E402
from tests._util import make_grid, rms_rel, max_rel, seeded  # noqa: E402


def main():
    from paradis_model_amd import ops
    print("grid mode | rms vs fp64: y(gpu cpu32) gf(gpu cpu32) gu(gpu cpu32) gv(gpu cpu32) | max y gpu-vs-cpu32")
    for H, W, poles, mode in [(32, 64, False, "bicubic"), (33, 64, True, "bilinear"),
                              (128, 256, False, "bicubic"), (65, 130, True, "bicubic")]:
        B, K = 2, 4
        _, lg, og = make_grid(H, W, poles)
        f, u, v, ct = (seeded(50 + i, B, K, H, W) for i in range(4))
        geo = O.GridGeometry(lg.double(), og.double())
        fd, ud, vd = (t.double().requires_grad_(True) for t in (f, u, v))
        yr = O.sl_advect_core(fd, ud, vd, 0.196887, geo, mode)
        yr.backward(ct.double())
        f32, u32, v32 = (t.clone().requires_grad_(True) for t in (f, u, v))
        y32 = O.sl_advect_core(f32, u32, v32, 0.196887, O.GridGeometry(lg, og), mode)
        y32.backward(ct)
        geom = ops.AdvectGeometry(lg, og)
        fg, ug, vg = (t.cuda().requires_grad_(True) for t in (f, u, v))
        y = ops.sl_advect(fg, ug, vg, geom, 0.196887, mode)
        y.backward(ct.cuda())
        r = lambda a, b: rms_rel(a.detach().cpu(), b.detach())
        print(f"{H}x{W} {mode:8s} | {r(y, yr):.2e} {r(y32, yr):.2e} | {r(fg.grad, fd.grad):.2e} "
              f"{r(f32.grad, fd.grad):.2e} | {r(ug.grad, ud.grad):.2e} {r(u32.grad, ud.grad):.2e} | "
              f"{r(vg.grad, vd.grad):.2e} {r(v32.grad, vd.grad):.2e} | {max_rel(y.detach().cpu(), y32.detach()):.2e}")
        # distribution of the gv error: a handful of ill-conditioned points (|sin lat_d| -> 1, where
        # d asin = 1/sqrt(1-s^2) amplifies one ulp of s) carries the rms
        for name, g32, gg, gr in (("gu", u32.grad, ug.grad.cpu(), ud.grad), ("gv", v32.grad, vg.grad.cpu(), vd.grad)):
            sc = float(gr.abs().max())
            eg = ((gg.double() - gr).abs() / sc).flatten()
            ec = ((g32.double() - gr).abs() / sc).flatten()
            q = torch.tensor([0.5, 0.99, 0.999], dtype=torch.float64)
            print(f"    {name} |err|/max: gpu median/99%/99.9%/max = "
                  + " ".join(f"{x:.1e}" for x in torch.quantile(eg, q).tolist()) + f" {float(eg.max()):.1e}"
                  + "   cpu32 = " + " ".join(f"{x:.1e}" for x in torch.quantile(ec, q).tolist())
                  + f" {float(ec.max()):.1e}")


if __name__ == "__main__":
    main()
