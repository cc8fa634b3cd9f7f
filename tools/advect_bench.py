#!/usr/bin/env python3
"""Diagnostic: the advection gather kernels in isolation against their algorithmic bytes
(16 B/point forward, 28 B/point backward) at the BASELINE shapes.
    python tools/advect_bench.py [vel_scale ...]      (default scales 0.05 0.5; ADVECT_BENCH_ONLY=32x64 for one shape)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids


def timeit(fn, n=20):
    for _ in range(10):    # the chip needs a few hundred ms of work to reach its steady clock
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    scales = [float(a) for a in sys.argv[1:]] or [0.05, 0.5]
    x = torch.randn(64 << 20, device="cuda")
    for _ in range(3000):     # ~2 s of warm-up: measurements taken cold read 20-30 % slow
        x = x * 1.0001
    for (B, K, H, W, poles) in ((32, 768, 32, 64, False), (8, 768, 128, 256, False), (1, 768, 721, 1440, True)):
        if os.environ.get("ADVECT_BENCH_ONLY", f"{H}x{W}") != f"{H}x{W}":
            continue
        _, lg, og = make_grids(H, W, poles)
        geom = ops.AdvectGeometry(lg, og)
        pts = B * K * H * W
        for scale in scales + (["smooth"] if os.environ.get("ADVECT_BENCH_SMOOTH", "1") == "1" else []):
            f = torch.randn(B, K, H, W, device="cuda", requires_grad=True)
            if scale == "smooth":
                # what a trained model's velocity fields look like next to white noise: a zonal jet of ~12 cells per
                # layer step (u dt = 12 cells) with smooth perturbations of a few cells, |v dt| ~ 2 cells
                cells = 2 * 3.14159265 / W          # radians per cell
                base = torch.randn(B, 2 * K, H // 8 + 1, W // 8 + 1, device="cuda")
                sm = torch.nn.functional.interpolate(base, size=(H, W), mode="bicubic", align_corners=False)
                vel = sm * (3 * cells / (0.196887 / 8))
                vel[:, :K] += 12 * cells / (0.196887 / 8) * torch.cos(lg.cuda())[None, None] ** 2
                vel[:, K:] *= 0.5
                vel = vel.contiguous().requires_grad_(True)
            else:
                vel = (torch.randn(B, 2 * K, H, W, device="cuda") * scale).requires_grad_(True)
            go = torch.randn(B, K, H, W, device="cuda")
            for mode in ("bicubic", "bilinear"):
                for name, flags in (("auto", None), ("generic", ops.advect_flags(generic=True))):
                    if name == "generic" and not (W == 64):
                        continue
                    y = None

                    def fwd():
                        nonlocal y
                        y = ops.sl_advect_vel(f, vel, geom, 0.196887 / 8, mode, flags=flags)

                    def fwd_nograd():     # without autograd bookkeeping the host keeps ahead of the kernel
                        with torch.no_grad():
                            ops.sl_advect_vel(f, vel, geom, 0.196887 / 8, mode, flags=flags)

                    def bwd():
                        f.grad = None; vel.grad = None
                        y.backward(go, retain_graph=True)
                    tf = timeit(fwd_nograd)
                    fwd()
                    tb = timeit(bwd)
                    print(f"{H}x{W} B={B} {mode:8s} {name:7s} vel_scale={scale}: fwd {1e3 * tf:7.1f} us "
                          f"{16 * pts / tf / 1e6:6.0f} GB/s ({16 * pts / tf / 8e9 * 100:4.1f} % of 8 TB/s)   "
                          f"bwd {1e3 * tb:7.1f} us {28 * pts / tb / 1e6:6.0f} GB/s ({28 * pts / tb / 8e9 * 100:4.1f} %)")


if __name__ == "__main__":
    main()
