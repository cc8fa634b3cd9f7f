#!/usr/bin/env python3
"""Diagnostic: the split weight-gradient GEMM alone at a layer shape and grid - time per launch, and (under
`rocprofv3 --pmc FETCH_SIZE`, tools/wgrad_traffic.sh) the bytes its L2 misses pull in per launch against the
operands' 4 (Co + Ci) N bytes.     python tools/wgrad_traffic.py [H W B [Co Ci]]     PARADIS_HIP_LIB = A/B builds"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd._lib import dptr, lib, stream_ptr

a = [int(x) for x in sys.argv[1:]]
H, W, B = a[:3] if len(a) >= 3 else (128, 256, 8)
shapes = [tuple(a[3:5])] if len(a) >= 5 else [(1024, 1024), (896, 1152), (1536, 384)]
P, st = H * W, stream_ptr()
w = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(40):
    w @ w
for Co, Ci in shapes:
    x = torch.randn(B, Ci, P, device="cuda")
    dy = torch.randn(B, Co, P, device="cuda")
    dw = torch.empty(Co, Ci, device="cuda")
    ws = torch.empty(lib.paradis_pw_gemm_wgrad_ws_bytes(B, Co, Ci, P), dtype=torch.uint8, device="cuda")
    run = lambda: lib.paradis_pw_gemm_wgrad(dptr(dy), dptr(x), dptr(dw), None, B, Co, Ci, P, Co * P, Ci * P, 3, None, None,
                                            dptr(ws), st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    ref = torch.einsum("bop,bcp->oc", dy[:1].double(), x[:1].double())
    lib.paradis_pw_gemm_wgrad(dptr(dy), dptr(x), dptr(dw), None, 1, Co, Ci, P, Co * P, Ci * P, 3, None, None, dptr(ws), st)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    print(f"wgrad {Co}x{Ci} N={B}x{P}: {us:8.1f} us = {2.0 * B * Co * Ci * P / us / 1e6:6.1f} TF fp32-eq; operands "
          f"{4.0 * (Co + Ci) * B * P / 1e9:.2f} GB; slabs {lib.paradis_pw_gemm_wgrad_slabs(B, Co, Ci, P)}; err {err:.1e}", flush=True)
