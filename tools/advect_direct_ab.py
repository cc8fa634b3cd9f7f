#!/usr/bin/env python3
"""Diagnostic (verdict r5 item 5): the forward gather of the large grids through the LDS ring (strips, shipped) against
the no-window variant that takes every tap from L2 with a plane per XCD (development library only - `make -C
paradis_model_amd/csrc dev` - PARADIS_ADVECT_DIRECT=4|8 rows per workgroup; the knob is read once per process, so one
process per arm).  Result: profiles/r06_advect_direct.txt.
    PARADIS_HIP_LIB=paradis_model_amd/libparadis_hip_dev.so PARADIS_ADVECT_DIRECT=<0|4|8> python tools/advect_direct_ab.py <ref.pt>
    (the first arm writes its outputs to ref.pt, later arms compare with them)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops
from paradis_model_amd.harness import make_grids

DT = 0.196887 / 8


def timeit(fn, n=20):
    for _ in range(10):
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
    ref_path = sys.argv[1]
    arm = os.environ.get("PARADIS_ADVECT_DIRECT", "0")
    ref = torch.load(ref_path) if os.path.exists(ref_path) else None
    outs = {}
    x = torch.randn(64 << 20, device="cuda")
    for _ in range(3000):
        x = x * 1.0001
    for (B, K, H, W, poles) in ((8, 768, 128, 256, False), (1, 768, 721, 1440, True)):
        _, lg, og = make_grids(H, W, poles)
        geom = ops.AdvectGeometry(lg, og)
        pts = B * K * H * W
        g = torch.Generator(device="cuda").manual_seed(7)
        f = torch.randn(B, K, H, W, device="cuda", generator=g)
        for scale in (0.05, 0.3, 1.0, "smooth"):
            if scale == "smooth":
                cells = 2 * 3.14159265 / W
                base = torch.randn(B, 2 * K, H // 8 + 1, W // 8 + 1, device="cuda", generator=g)
                vel = torch.nn.functional.interpolate(base, size=(H, W), mode="bicubic", align_corners=False) * (3 * cells / DT)
                vel[:, :K] += 12 * cells / DT * torch.cos(lg.cuda())[None, None] ** 2
                vel[:, K:] *= 0.5
                vel = vel.contiguous()
            else:
                vel = torch.randn(B, 2 * K, H, W, device="cuda", generator=g) * scale
            for mode in ("bicubic", "bilinear"):
                with torch.no_grad():
                    t = timeit(lambda: ops.sl_advect_vel(f, vel, geom, DT, mode))
                    y = ops.sl_advect_vel(f, vel, geom, DT, mode)
                key = f"{H}x{W}/{scale}/{mode}"
                chk = ""
                if ref is not None:
                    d = (y.cpu() - ref[key]).abs().max().item()
                    chk = f"  max|y - y_first_arm| = {d:.3g}"
                else:
                    outs[key] = y.cpu()
                print(f"direct={arm} {H}x{W} B={B} {mode:8s} vel={scale}: fwd {1e3 * t:7.1f} us = {16 * pts / t / 8e9 * 100:4.1f} % of 8 TB/s{chk}",
                      flush=True)
    if ref is None:
        torch.save(outs, ref_path)


if __name__ == "__main__":
    main()
