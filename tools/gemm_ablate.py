#!/usr/bin/env python3
"""Diagnostic: time the f16x2 forward GEMM (1024x1024 weights, 32 x 2048 points) of ONE library build; used with
ablation builds under build/variants/ (scratch copies of gemm.hip with one piece of the k-loop removed - their
results are wrong by construction, only the timing matters), rotated by the calling shell loop:
    for r in 1 2 3; do for n in base nob ...; do PARADIS_HIP_LIB=build/variants/lib_$n.so python tools/gemm_ablate.py $n; done; done"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd._lib import dptr, lib, stream_ptr

B, P, Co, Ci = 32, 2048, 1024, 1024
w = torch.randn(Co, Ci, device="cuda") / 32
x = torch.randn(B, Ci, P, device="cuda")
y = torch.empty(B, Co, P, device="cuda")
st = stream_ptr()
wsp = torch.empty(lib.paradis_pw_gemm_split_bytes(Co, Ci, 2), dtype=torch.uint8, device="cuda")
lib.paradis_pw_gemm_split_weights(dptr(w), Co, Ci, 0, 2, dptr(wsp), st)
xa = torch.empty(1024, dtype=torch.int32, device="cuda")
lib.paradis_amax_partials(dptr(x), B, Ci * P, Ci * P, dptr(xa), st)
run = lambda: lib.paradis_pw_gemm_fwd(dptr(w), None, dptr(wsp), 2, dptr(xa), dptr(x), None, None, None, None, 0, None,
                                      dptr(y), None, B, Co, Ci, P, Ci * P, 0, Co * P, 0, st)
for _ in range(30):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record(); torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'lib':10s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
