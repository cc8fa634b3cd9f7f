#!/usr/bin/env python3
"""Diagnostic: tiled advection kernels (128x256, 721x1440) vs the LDS window halo.
Usage on the GPU box: python tools/advect_halo_sweep.py [vel_scale]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops                     # noqa: E402
from paradis_model_amd._lib import lib               # noqa: E402
from paradis_model_amd.harness import make_grids     # noqa: E402


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    for B, K, H, W, poles in ((8, 768, 128, 256, False),):
        _, lg, og = make_grids(H, W, poles)
        geom = ops.AdvectGeometry(lg, og)
        f = torch.randn(B, K, H, W, device="cuda", requires_grad=True)
        vel = (torch.randn(B, 2 * K, H, W, device="cuda") * scale).requires_grad_(True)
        go = torch.randn(B, K, H, W, device="cuda")
        pts = B * K * H * W
        for hf, hb in ((4, 4), (6, 6), (8, 10), (12, 10), (16, 12), (24, 14)):
            flags = ops.advect_flags(halo=hf, halo_bwd=hb)
            y = None

            def fwd():
                nonlocal y
                y = ops.sl_advect(f, vel[:, :K], vel[:, K:], geom, 0.196887, "bicubic", flags=flags)

            def bwd():
                f.grad = None; vel.grad = None
                y.backward(go, retain_graph=True)
            tf = timeit(fwd)
            tb = timeit(bwd)
            print(f"{H}x{W} B={B} vel_scale={scale}: halo fwd {hf:2d} -> {tf:7.3f} ms ({16 * pts / tf / 1e6:6.0f} GB/s)   "
                  f"halo bwd {hb:2d} -> {tb:7.3f} ms ({28 * pts / tb / 1e6:6.0f} GB/s)")


if __name__ == "__main__":
    main()
