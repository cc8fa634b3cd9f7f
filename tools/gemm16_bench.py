#!/usr/bin/env python3
"""bf16-mixed pointwise GEMMs, isolated launches through the C ABI: fp32-stored (k32 kernel) against bf16-stored (LDS-DMA +
transposed reads) operands / outputs, per layer shape, N = 32 x 2048 (32x64 grid, B = 32).  K = 32 rows of the same launch
separate the epilogue (independent of K) from the k-loop.
  python tools/gemm16_bench.py [--shapes 896x896,1024x768] [--reps 20]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from paradis_model_amd import ops  # noqa: E402
from paradis_model_amd._lib import dptr, lib, stream_ptr  # noqa: E402

BF = ops.GEMM_BF16


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def image(w, transpose):
    Co, Ci = w.shape
    n = lib.paradis_pw_gemm_split_bytes(Ci, Co, BF) if transpose else lib.paradis_pw_gemm_split_bytes(Co, Ci, BF)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    assert lib.paradis_pw_gemm_split_weights(dptr(w), Co, Ci, 1 if transpose else 0, BF, dptr(out), stream_ptr()) == 0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="896x896,1024x896,896x1152,1024x1024,384x1024,1536x384,768x1024,1024x768")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--P", type=int, default=2048)
    a = ap.parse_args()
    B, P = a.B, a.P
    print("%-10s %-34s %9s %8s" % ("Co x Ci", "variant", "us", "TF"))
    for sh in a.shapes.split(","):
        Co, Ci = (int(v) for v in sh.split("x"))
        for K in (Ci, 32):
            w = (torch.randn(Co, K, device="cuda") / K ** 0.5)
            img = image(w, False)
            x32 = torch.randn(B, K, P, device="cuda")
            x16 = x32.to(torch.bfloat16)
            bias = torch.zeros(Co, device="cuda")
            y32, z32 = torch.empty(B, Co, P, device="cuda"), torch.empty(B, Co, P, device="cuda")
            y16, z16 = y32.to(torch.bfloat16), z32.to(torch.bfloat16)
            fl = 2.0 * B * Co * K * P

            def fwd(x, y, z, io, act=1):
                return lambda: lib.paradis_pw_gemm_fwd16(dptr(img), dptr(x), dptr(bias), None, None, None, 0, None, None,
                                                         dptr(y), dptr(z) if z is not None else None, B, Co, K, P, K * P, 0,
                                                         Co * P, act, io, stream_ptr())
            rows = [("fwd x32 -> y32,z32 (k32)", fwd(x32, y32, z32, 0)),
                    ("fwd x32 -> y16,z16 (k32)", fwd(x32, y16, z16, 2)),
                    ("fwd x16 -> y16,z16 (dma+tr)", fwd(x16, y16, z16, 3)),
                    ("fwd x16 -> y32 no act/z", fwd(x16, y32, None, 1, act=0)),
                    ("fwd x32 -> y32 no act/z (k32)", fwd(x32, y32, None, 0, act=0))]
            for name, fn in rows:
                us = timed(fn, a.reps)
                print("%-10s %-34s %9.1f %8.1f%s" % (f"{Co}x{K}", name, us, fl / us / 1e6, "   (epilogue probe)" if K != Ci else ""))
        # weight gradient
        dy32 = torch.randn(B, Co, P, device="cuda")
        xx32 = torch.randn(B, Ci, P, device="cuda")
        dy16, xx16 = dy32.to(torch.bfloat16), xx32.to(torch.bfloat16)
        gw = torch.empty(Co, Ci, device="cuda")
        ws = torch.empty(lib.paradis_pw_gemm_wgrad_ws_bytes(B, Co, Ci, P) // 4 + 64, device="cuda")
        fl = 2.0 * B * Co * Ci * P
        for name, (dy, xx, io) in (("wgrad dy32 x32", (dy32, xx32, 0)), ("wgrad dy16 x16", (dy16, xx16, 9)),
                                   ("wgrad dy32 x16", (dy32, xx16, 1)), ("wgrad dy16 x32", (dy16, xx32, 8))):
            if io == 0:
                fn = lambda dy=dy, xx=xx: lib.paradis_pw_gemm_wgrad(dptr(dy), dptr(xx), dptr(gw), None, B, Co, Ci, P, Co * P, Ci * P,
                                                                    BF, None, None, dptr(ws), stream_ptr())
            else:
                fn = lambda dy=dy, xx=xx, io=io: lib.paradis_pw_gemm_wgrad16(dptr(dy), dptr(xx), dptr(gw), None, B, Co, Ci, P, Co * P,
                                                                             Ci * P, io, dptr(ws), stream_ptr())
            us = timed(fn, a.reps)
            print("%-10s %-34s %9.1f %8.1f" % (f"{Co}x{Ci}", name, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
