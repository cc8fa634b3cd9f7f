import os
# needs the development build of the library (make -C paradis_model_amd/csrc dev): the shipped one exports
# no paradis_debug_set_* tunables
os.environ.setdefault("PARADIS_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                      "paradis_model_amd", "libparadis_hip_dev.so"))

#!/usr/bin/env python3
"""Diagnostic: A/B the FP32-MFMA pointwise GEMM tunables (k-tile depth, resident workgroups per CU)
over the GEMM shapes of the default PARADIS layer, interleaved in one process (HIP events)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import _lib  # noqa: E402
from paradis_model_amd._lib import dptr, lib, stream_ptr  # noqa: E402

B, P = 32, 2048
ACCURACY = os.environ.get("GEMM_ACCURACY", "1") == "1"
SHAPES = [(1024, 186), (384, 1024), (1536, 384), (768, 1024), (1024, 768), (1024, 1024), (896, 1152),
          (896, 896), (1024, 896), (768, 768), (97, 768)]   # (Co, Ci)
# (bk, workgroups/CU of the register-staged kernel, LDS-DMA ring depth [0 = off], bf16-split [0/1])
CONFIGS = [(16, 4, 3, 0), (16, 4, 3, 1)] if len(sys.argv) < 2 else \
    [tuple(map(int, a.split(","))) for a in sys.argv[1:]]
CONFIGS = [(tuple(c) + (0, 0))[:4] for c in CONFIGS]
lib.paradis_debug_set_gemm_dma.argtypes = [ctypes.c_int]
lib.paradis_debug_set_wgrad_dma.argtypes = [ctypes.c_int]
lib.paradis_debug_set_gemm.argtypes = [ctypes.c_int, ctypes.c_int]
lib.paradis_debug_set_gemm_stagger.argtypes = [ctypes.c_int]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    st = stream_ptr()
    tot = {c: {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0} for c in CONFIGS}
    for (Co, Ci) in SHAPES:
        w = torch.randn(Co, Ci, device="cuda") * Ci ** -0.5
        wt = w.t().contiguous()
        x = torch.randn(B, Ci, P, device="cuda")
        dy = torch.randn(B, Co, P, device="cuda")
        y = torch.empty(B, Co, P, device="cuda")
        dx = torch.empty(B, Ci, P, device="cuda")
        dw = torch.empty(Co, Ci, device="cuda")
        ws = torch.empty(64 << 20, device="cuda")
        flops = 2.0 * B * Co * Ci * P
        line = f"Co={Co:5d} Ci={Ci:5d} |"
        ref = None
        for cfg in CONFIGS:
            lib.paradis_debug_set_gemm(cfg[0], cfg[1])
            lib.paradis_debug_set_gemm_dma(cfg[2])
            lib.paradis_debug_set_wgrad_dma(cfg[2] if cfg[2] <= 3 else 3)
            use_t = cfg[2] >= 2
            wsp = wtsp = None
            if cfg[3]:
                wsp = torch.empty(lib.paradis_pw_gemm_split_bytes(Co, Ci, 3), dtype=torch.uint8, device="cuda")
                wtsp = torch.empty(lib.paradis_pw_gemm_split_bytes(Ci, Co, 3), dtype=torch.uint8, device="cuda")
                lib.paradis_pw_gemm_split_weights(dptr(w), Co, Ci, 0, 3, dptr(wsp), st)
                lib.paradis_pw_gemm_split_weights(dptr(w), Co, Ci, 1, 3, dptr(wtsp), st)
            t = {}
            t["fwd"] = timeit(lambda: lib.paradis_pw_gemm_fwd(dptr(w), dptr(wt) if use_t else None, dptr(wsp), 3 if wsp is not None else 0, None, dptr(x), None, None, None, None, 0, None, dptr(y), None, B, Co, Ci, P, Ci * P, 0, Co * P, 0, st))
            t["dgrad"] = timeit(lambda: lib.paradis_pw_gemm_dgrad(dptr(w), dptr(wtsp), 3 if wtsp is not None else 0, None, dptr(dy), None, None, dptr(dx), B, Co, Ci, P, Co * P, 0, 0, Ci * P, 0, st))
            t["wgrad"] = timeit(lambda: lib.paradis_pw_gemm_wgrad(dptr(dy), dptr(x), dptr(dw), None, B, Co, Ci, P, Co * P, Ci * P, 3 if cfg[3] else 0, None, None, dptr(ws), st))
            if ACCURACY:   # max error / max |exact| against fp64 on two samples
                if ref is None:
                    xd, dyd, wd = x[:2].double(), dy[:2].double(), w.double()
                    ref = (wd @ xd, wd.t() @ dyd, torch.einsum("bmp,bkp->mk", dy.double(), x.double()))
                errs = [float(((a.double() - r).abs().max() / r.abs().max()))
                        for a, r in ((y[:2], ref[0]), (dx[:2], ref[1]), (dw, ref[2]))]
                line += "  err " + " ".join(f"{e:.1e}" for e in errs)
            for k in t:
                tot[cfg][k] += t[k]
            line += "  bk%d/wg%d/dma%d/split%d: " % cfg + " ".join(f"{k[0]}{flops / t[k] / 1e6:6.1f}" for k in ("fwd", "dgrad", "wgrad"))
        print(line, flush=True)
    for cfg in CONFIGS:
        print("total us bk%d/wg%d/dma%d/split%d:" % cfg, {k: round(v) for k, v in tot[cfg].items()}, "sum", round(sum(tot[cfg].values())))


if __name__ == "__main__":
    main()
