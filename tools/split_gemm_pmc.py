#!/usr/bin/env python3
"""Diagnostic: 100 launches of a split forward GEMM (1024x1024 weights, 32 x 2048 points; argv[1] = f16x2 | bf16x3)
for rocprofv3 --pmc passes (PMC_SCRIPT=tools/split_gemm_pmc.py bash tools/advect_pmc.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd._lib import dptr, lib, stream_ptr  # noqa: E402

B, P, Co, Ci = 32, 2048, 1024, 1024
SCHEME = {"f16x2": 2, "bf16x3": 3}[sys.argv[1] if len(sys.argv) > 1 else "f16x2"]
w = torch.randn(Co, Ci, device="cuda") / 32
x = torch.randn(B, Ci, P, device="cuda")
y = torch.empty(B, Co, P, device="cuda")
wsp = torch.empty(lib.paradis_pw_gemm_split_bytes(Co, Ci, SCHEME), dtype=torch.uint8, device="cuda")
st = stream_ptr()
lib.paradis_pw_gemm_split_weights(dptr(w), Co, Ci, 0, SCHEME, dptr(wsp), st)
xa = torch.empty(1024, dtype=torch.int32, device="cuda")
lib.paradis_amax_partials(dptr(x), B, Ci * P, Ci * P, dptr(xa), st)
for _ in range(100):
    assert lib.paradis_pw_gemm_fwd(dptr(w), None, dptr(wsp), SCHEME, dptr(xa), dptr(x), None, None, None, None, 0, None, dptr(y),
                                   None, B, Co, Ci, P, Ci * P, 0, Co * P, 0, st) == 0
torch.cuda.synchronize()
