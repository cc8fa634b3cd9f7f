#!/usr/bin/env python3
"""Diagnostic: what the epilogue of the fp32-width (bf16x3) forward GEMM costs - the same launch with no activation, with
SiLU, with SiLU + the saved pre-activation, and with a residual, at the full K and at K = 32 (epilogue only).
    python tools/gemm_epilogue_probe.py [CoxCi ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops

B, H, W = 32, 32, 64


def timeit(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 1024), (896, 896)]
    big = torch.randn(64 << 20, device="cuda")
    for _ in range(2000):
        big = big * 1.0001
    for Co, Ci in shapes:
        for K in (Ci, 32):
            x = torch.randn(B, K, H, W, device="cuda")
            w = torch.nn.Parameter(torch.randn(Co, K, 1, 1, device="cuda") / K ** 0.5)
            b = torch.nn.Parameter(torch.zeros(Co, device="cuda"))
            res = torch.randn(B, Co, H, W, device="cuda")
            row = []
            with torch.no_grad():
                row.append(("no act", timeit(lambda: ops.pointwise(x, w, b))))
                row.append(("SiLU", timeit(lambda: ops.pointwise(x, w, b, act="SiLU"))))
                row.append(("residual", timeit(lambda: ops.pointwise(x, w, b, residual=res))))
            row.append(("SiLU + z", timeit(lambda: ops.pointwise(x, w, b, act="SiLU"))))       # grad mode: z is saved
            print("%4d x %4d  " % (Co, K) + "   ".join("%s %.1f us" % r for r in row), flush=True)


if __name__ == "__main__":
    main()
