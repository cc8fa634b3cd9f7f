import os
# needs the development build of the library (make -C paradis_model_amd/csrc dev): the shipped one exports
# no paradis_debug_set_* tunables
os.environ.setdefault("PARADIS_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                      "paradis_model_amd", "libparadis_hip_dev.so"))

#!/usr/bin/env python3
"""Diagnostic: time ablated builds of the pointwise GEMM (results are WRONG by construction; only
the timing matters) to attribute the matrix-pipe idle time."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "paradis_model_amd", "csrc")
OUT = os.path.join(ROOT, "build", "variants")
VARIANTS = {"base": [], "base_dma": [], "dma_no_barrier": ["-DDMA_NO_BARRIER"], "dma_no_issue": ["-DDMA_NO_ISSUE"],
            "dma_no_ldsread": ["-DDMA_NO_LDSREAD"], "dma_no_epilogue": ["-DDMA_NO_EPILOGUE"],
            "dma_mfma_only": ["-DDMA_NO_BARRIER", "-DDMA_NO_ISSUE", "-DDMA_NO_LDSREAD", "-DDMA_NO_EPILOGUE"],
            "dma_no_bar_issue": ["-DDMA_NO_BARRIER", "-DDMA_NO_ISSUE"],
            "no_stage": ["-DGEMM_NO_STAGE"],
            "no_stage_no_barrier": ["-DGEMM_NO_STAGE", "-DGEMM_NO_BARRIER"],
            "mfma_only": ["-DGEMM_NO_STAGE", "-DGEMM_NO_BARRIER", "-DGEMM_NO_LDSREAD"],
            "no_barrier": ["-DGEMM_NO_BARRIER"], "no_ldsstore": ["-DGEMM_NO_LDSSTORE"],
            "no_gload": ["-DGEMM_NO_GLOAD"], "unguarded": ["-DGEMM_UNGUARDED"],
            "unguarded_no_ldsstore": ["-DGEMM_UNGUARDED", "-DGEMM_NO_LDSSTORE"],
            "split_base": [], "split_no_fetch": ["-DSPLIT_NO_FETCH"], "split_no_dma": ["-DSPLIT_NO_DMA"],
            "split_no_ldsread": ["-DSPLIT_NO_LDSREAD"], "split_no_store": ["-DSPLIT_NO_STORE"],
            "split_no_barrier": ["-DSPLIT_NO_BARRIER"], "split_no_epilogue": ["-DSPLIT_NO_EPILOGUE"],
            "split_no_interleave": ["-DSPLIT_NO_INTERLEAVE"],
            "split_no_global": ["-DSPLIT_NO_FETCH", "-DSPLIT_NO_DMA"],
            "split_no_global_store": ["-DSPLIT_NO_FETCH", "-DSPLIT_NO_DMA", "-DSPLIT_NO_STORE"],
            "split_mfma_only": ["-DSPLIT_NO_FETCH", "-DSPLIT_NO_DMA", "-DSPLIT_NO_STORE", "-DSPLIT_NO_LDSREAD",
                                "-DSPLIT_NO_BARRIER", "-DSPLIT_NO_EPILOGUE"]}


def build():
    os.makedirs(OUT, exist_ok=True)
    only = os.environ.get("GEMM_VARIANTS")
    for name, flags in VARIANTS.items():
        if only and not (name in only.split(",") or (only == "split" and name.startswith("split_"))):
            continue
        so = os.path.join(OUT, f"libgemm_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17",
               "-munsafe-fp-atomics", "-fno-slp-vectorize", "-shared", *flags, os.path.join(CSRC, "gemm.hip"),
               os.path.join(CSRC, "error.hip"), os.path.join(CSRC, "misc.hip"), "-o", so]
        subprocess.run(cmd, check=True)


def main():
    if not torch.cuda.is_available():
        build()
        print("built")
        return
    from paradis_model_amd import _lib
    B, P, Co, Ci = 32, 2048, 1024, 1024
    w = torch.randn(Co, Ci, device="cuda") / 32
    x = torch.randn(B, Ci, P, device="cuda")
    y = torch.empty(B, Co, P, device="cuda")
    wt = w.t().contiguous()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    flops = 2.0 * B * Co * Ci * P
    libs = {}
    only = os.environ.get("GEMM_VARIANTS")
    names = [n for n in VARIANTS if not only or n in only.split(",") or (only == "split" and n.startswith("split_"))]
    wsp = torch.empty(_lib.lib.paradis_pw_gemm_split_bytes(Co, Ci), dtype=torch.uint8, device="cuda")
    _lib.lib.paradis_pw_gemm_split_weights(p(w), Co, Ci, 0, p(wsp), st)
    for name in names:
        L = ctypes.CDLL(os.path.join(OUT, f"libgemm_{name}.so"))
        L.paradis_pw_gemm_fwd.argtypes = _lib.SIGNATURES["paradis_pw_gemm_fwd"][1]
        libs[name] = L
    occ = [int(a) for a in sys.argv[1:]] or [4]
    for name, L in libs.items():
        L.paradis_debug_set_gemm.argtypes = [ctypes.c_int, ctypes.c_int]
    rounds = int(os.environ.get("GEMM_ROUNDS", "8"))
    for wg in occ:
        for L in libs.values():
            L.paradis_debug_set_gemm(16, wg)
        print("workgroups per CU:", wg, "(median of", rounds, "alternating rounds of 10 launches; the order inside a"
              " round is rotated: a variant's rate depends on what ran just before it)")
        sel = [n for n in libs if n in ("base", "no_stage", "mfma_only", "no_gload", "unguarded") or wg == 4
               or n.startswith("split_")]
        times = {n: [] for n in sel}
        for rnd in range(rounds + 1):
            order = sel[rnd % len(sel):] + sel[:rnd % len(sel)]
            for name in order:
                L = libs[name]
                # dma_* variants exercise the LDS-DMA kernel (transposed weights supplied), split_* the bf16-split
                # kernel (weight image supplied), the others the register-staged kernel
                wt_arg = p(wt) if name.startswith("dma_") or name == "base_dma" else None
                sp_arg = p(wsp) if name.startswith("split_") else None
                fn = lambda: L.paradis_pw_gemm_fwd(p(w), wt_arg, sp_arg, p(x), None, None, None, None, 0, None, p(y), None, B, Co, Ci, P, Ci * P, 0, Co * P, 0, st)
                assert fn() == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record(); torch.cuda.synchronize()
                if rnd:
                    times[name].append(e0.elapsed_time(e1) / 10 * 1e3)
        for name in sel:
            t = sorted(times[name])
            us = t[len(t) // 2]
            print(f"{name:22s} {us:8.1f} us  {flops / us / 1e6:6.1f} TF   (min {t[0]:.1f}, max {t[-1]:.1f})")


if __name__ == "__main__":
    main()
