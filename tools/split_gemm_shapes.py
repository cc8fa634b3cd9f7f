#!/usr/bin/env python3
"""Diagnostic: the split forward / data-gradient / weight-gradient GEMMs on every (Co, Ci) of the default layer
at B=32, P=2048 with the shipped library: TF/s (fp32-equivalent) and error against fp64.
    python tools/split_gemm_shapes.py [f16x2|bf16x3]      (default f16x2 = the scheme this tool was written for; amax passes timed separately)
  PARADIS_HIP_LIB
selects another build of the same ABI for an A/B (round 2: a warp-specialised 128x256 kernel - four
MFMA-only waves, four loader/splitter waves, four LDS stages - measured 178 TF where this one does 200)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd._lib import dptr, lib, stream_ptr

B, P = 32, 2048
SCHEME = {"f16x2": 2, "bf16x3": 3}[sys.argv[1] if len(sys.argv) > 1 else "f16x2"]
SHAPES = [(1024, 186), (384, 1024), (1536, 384), (768, 1024), (1024, 768), (1024, 1024), (896, 1152),
          (896, 896), (1024, 896), (768, 768), (97, 768)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    st = stream_ptr()
    x = torch.randn(64 << 20, device="cuda")
    for _ in range(2000):
        x = x * 1.0001
    tot = [0.0, 0.0, 0.0, 0.0]
    for (Co, Ci) in SHAPES:
        w = torch.randn(Co, Ci, device="cuda") * Ci ** -0.5
        x = torch.randn(B, Ci, P, device="cuda")
        dy = torch.randn(B, Co, P, device="cuda")
        y = torch.empty(B, Co, P, device="cuda")
        dx = torch.empty(B, Ci, P, device="cuda")
        wsp = torch.empty(lib.paradis_pw_gemm_split_bytes(Co, Ci, SCHEME), dtype=torch.uint8, device="cuda")
        wtsp = torch.empty(lib.paradis_pw_gemm_split_bytes(Ci, Co, SCHEME), dtype=torch.uint8, device="cuda")
        lib.paradis_pw_gemm_split_weights(dptr(w), Co, Ci, 0, SCHEME, dptr(wsp), st)
        lib.paradis_pw_gemm_split_weights(dptr(w), Co, Ci, 1, SCHEME, dptr(wtsp), st)
        xa = torch.empty(1024, dtype=torch.int32, device="cuda")
        da = torch.empty(1024, dtype=torch.int32, device="cuda")
        dw = torch.empty(Co, Ci, device="cuda")
        ws = torch.empty(lib.paradis_pw_gemm_wgrad_ws_bytes(B, Co, Ci, P), dtype=torch.uint8, device="cuda")
        ta = timeit(lambda: (lib.paradis_amax_partials(dptr(x), B, Ci * P, Ci * P, dptr(xa), st),
                             lib.paradis_amax_partials(dptr(dy), B, Co * P, Co * P, dptr(da), st)))
        flops = 2.0 * B * Co * Ci * P
        tf = timeit(lambda: lib.paradis_pw_gemm_fwd(dptr(w), None, dptr(wsp), SCHEME, dptr(xa), dptr(x), None, None, None,
                                                   None, 0, None, dptr(y), None, B, Co, Ci, P, Ci * P, 0, Co * P, 0, st))
        td = timeit(lambda: lib.paradis_pw_gemm_dgrad(dptr(w), dptr(wtsp), SCHEME, dptr(da), dptr(dy), None, None, dptr(dx),
                                                     B, Co, Ci, P, Co * P, 0, 0, Ci * P, 0, st))
        tw = timeit(lambda: lib.paradis_pw_gemm_wgrad(dptr(dy), dptr(x), dptr(dw), None, B, Co, Ci, P, Co * P, Ci * P,
                                                     SCHEME, dptr(da), dptr(xa), dptr(ws), st))
        wd = w.double()
        e1 = float((y[:2].double() - wd @ x[:2].double()).abs().max() / (wd @ x[:2].double()).abs().max())
        e2 = float((dx[:2].double() - wd.t() @ dy[:2].double()).abs().max() / (wd.t() @ dy[:2].double()).abs().max())
        e3 = float((dw.double() - torch.einsum("bop,bcp->oc", dy.double(), x.double())).abs().max()
                   / torch.einsum("bop,bcp->oc", dy[:1].double(), x[:1].double()).abs().max() / B ** 0.5)
        tot[0] += tf; tot[1] += td; tot[2] += tw; tot[3] += ta
        print(f"Co={Co:5d} Ci={Ci:5d}  fwd {tf:7.1f} us {flops / tf / 1e6:6.1f} TF  dgrad {td:7.1f} us {flops / td / 1e6:6.1f} TF"
              f"  wgrad {tw:7.1f} us {flops / tw / 1e6:6.1f} TF  amax(x)+amax(dy) {ta:6.1f} us   err {e1:.1e} {e2:.1e} {e3:.1e}",
              flush=True)
    print(f"sum fwd {tot[0]:.0f} us, dgrad {tot[1]:.0f} us, wgrad {tot[2]:.0f} us, amax passes {tot[3]:.0f} us")


if __name__ == "__main__":
    main()
