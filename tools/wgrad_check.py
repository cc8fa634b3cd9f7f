#!/usr/bin/env python3
"""bf16-mixed weight gradient: the selected kernel against an fp64 evaluation of the rounded operands (where are the bad entries?)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops
g = torch.Generator().manual_seed(1)
shapes = ((1, 896, 1024, 32, 64), (2, 1024, 1024, 32, 64), (1, 256, 256, 8, 16), (1, 256, 256, 4, 8), (3, 512, 768, 16, 32))
if len(sys.argv) > 1:
    shapes = tuple(tuple(int(v) for v in a.split(",")) for a in sys.argv[1:])
for (B, Co, Ci, H, W) in shapes:
    dz = torch.randn(B, Co, H, W, generator=g).cuda()
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    for d16, x16 in ((False, False), (True, False), (False, True), (True, True)):
        a = dz.to(torch.bfloat16) if d16 else dz
        b = x.to(torch.bfloat16) if x16 else x
        gw, _ = ops._pw_gemm_wgrad(a, b, False, None, None, ops.GEMM_BF16)
        ref = torch.einsum("bohw,bchw->oc", dz.to(torch.bfloat16).double(), x.to(torch.bfloat16).double())
        bad = ~torch.isfinite(gw)
        err = ((gw.double() - ref).abs() / ref.abs().max()).nan_to_num(9.0)
        rows = torch.nonzero(err.max(dim=1).values > 1e-3).flatten()
        cols = torch.nonzero(err.max(dim=0).values > 1e-3).flatten()
        print((B, Co, Ci, H, W), "dy16" if d16 else "dy32", "x16" if x16 else "x32", "slabs", ops.lib.paradis_pw_gemm_wgrad_slabs(B, Co, Ci, H * W), "nan", int(bad.sum()), "max rel err %.2e" % float(err.max()),
              "bad rows", rows[:6].tolist(), "..", rows[-3:].tolist() if len(rows) else [], "bad cols", cols[:6].tolist(), len(rows), len(cols))
