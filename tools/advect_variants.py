#!/usr/bin/env python3
"""Diagnostic: time ablated builds of the advection kernels (A/B in one process, HIP events).
Usage on the GPU box:  python tools/advect_variants.py   (builds side libraries under build/variants)"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "paradis_model_amd", "csrc")
OUT = os.path.join(ROOT, "build", "variants")
VARIANTS = {"base": [], "ieee_div": ["-DADV_IEEE_DIV"], "ocml_atan2": ["-DADV_OCML_ATAN2"],
            "ocml_sincos": ["-DADV_OCML_SINCOS"], "no_small": ["-DADV_NO_SMALL_ANGLE"], "no_atomic": ["-DADV_NO_ATOMIC"],
            "no_trig": ["-DADV_NO_TRIG"], "no_both": ["-DADV_NO_ATOMIC", "-DADV_NO_TRIG"],
            "cvt_i64": ["-DADV_CVT_I64"], "pf1": ["-DADV_PF=1"], "pf3": ["-DADV_PF=3"], "pf4": ["-DADV_PF=4"], "no_stage": ["-DADV_NO_STAGE"],
            "no_gather": ["-DADV_NO_GATHER"], "no_tables": ["-DADV_NO_TABLES"],
            "no_sgt": ["-DADV_NO_STAGE", "-DADV_NO_GATHER", "-DADV_NO_TABLES"],
            "no_sgtt": ["-DADV_NO_STAGE", "-DADV_NO_GATHER", "-DADV_NO_TABLES", "-DADV_NO_TRIG"]}
if os.environ.get("ADV_VARIANTS"):
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in os.environ["ADV_VARIANTS"].split(",")}


def build():
    os.makedirs(OUT, exist_ok=True)
    for name, flags in VARIANTS.items():
        so = os.path.join(OUT, f"libadv_{name}.so")
        # a flag "src=<file>" builds the variant from another source file (A/B of two revisions);
        # it is compiled from inside csrc/ so that its relative includes resolve
        src = os.path.join(CSRC, "advect.hip")
        for f in flags:
            if f.startswith("src="):
                src = os.path.join(ROOT, f[4:])
        flags = [f for f in flags if not f.startswith("src=")]
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17",
               "-munsafe-fp-atomics", "-ffp-contract=off", "-shared", "-I", CSRC, *flags,
               src, os.path.join(CSRC, "error.hip"), "-o", so]
        subprocess.run(cmd, check=True)


def main():
    build()
    if not torch.cuda.is_available():
        print("built variants; no GPU here")
        return
    from paradis_model_amd import _lib
    B, K, H, W = 32, 768, 32, 64
    from paradis_model_amd.harness import make_grids
    from paradis_model_amd.ops import AdvectGeometry
    _, lg, og = make_grids(H, W, False)
    geom = AdvectGeometry(lg, og)
    sl, cl, lo = geom.tables("cuda")
    f = torch.randn(B, K, H, W, device="cuda")
    vel = torch.randn(B, 2 * K, H, W, device="cuda")
    u, v = vel[:, :K], vel[:, K:]
    go = torch.randn(B, K, H, W, device="cuda")
    out = torch.empty_like(f); gf = torch.empty_like(f); guv = torch.empty_like(vel)
    ws = torch.empty(B * K * 4 + 64, device="cuda")
    P = K * H * W
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    libs = {}
    for name in VARIANTS:
        L = ctypes.CDLL(os.path.join(OUT, f"libadv_{name}.so"))
        L.paradis_sl_advect_fwd.argtypes = _lib.SIGNATURES["paradis_sl_advect_fwd"][1]
        L.paradis_sl_advect_bwd.argtypes = _lib.SIGNATURES["paradis_sl_advect_bwd"][1]
        libs[name] = L
    def fwd(L):
        return L.paradis_sl_advect_fwd(p(f), p(u), p(v), p(out), p(sl), p(cl), p(lo), B, K, H, W, P, 2 * P, P,
                                       0.196887, geom.min_lat, geom.min_lon, geom.d_lat, geom.d_lon, 2, p(ws), st)
    def bwd(L):
        return L.paradis_sl_advect_bwd(p(go), p(f), p(u), p(v), p(gf), p(guv[:, :K]), p(guv[:, K:]), p(sl), p(cl),
                                       p(lo), B, K, H, W, P, P, 2 * P, P, 2 * P, 0.196887, geom.min_lat,
                                       geom.min_lon, geom.d_lat, geom.d_lon, 2, p(ws), st)
    for scale in (1.0,):
        if scale == "smooth":   # spatially smooth velocity: neighbouring points are displaced alike
            yy = torch.linspace(0, 6.28, H, device="cuda").view(1, 1, H, 1)
            xx = torch.linspace(0, 6.28, W, device="cuda").view(1, 1, 1, W)
            ph = torch.rand(B, 2 * K, 1, 1, device="cuda") * 6.28
            vel.copy_(2.0 * torch.sin(yy + ph) * torch.cos(xx * 2 + ph))
        else:
            vel.normal_().mul_(scale)
        for kind, fn in (("fwd", fwd), ("bwd", bwd)):
            best = {name: 1e9 for name in libs}
            for rnd in range(5):   # interleaved rounds, best-of: box clocks drift by a few percent
                for name, L in libs.items():
                    assert fn(L) == 0
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn(L)
                    e1.record()
                    torch.cuda.synchronize()
                    best[name] = min(best[name], e0.elapsed_time(e1) / 20 * 1e3)
            for name in libs:
                print(f"vel_scale={scale} {kind} {name:12s} {best[name]:9.1f} us")


if __name__ == "__main__":
    main()
