#!/usr/bin/env python3
"""Diagnostic: time the ChannelNorm backward with/without the per-channel shuffle reduction."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "paradis_model_amd", "csrc"); OUT = os.path.join(ROOT, "build", "variants")
VARIANTS = {"base": [], "no_shuffle": ["-DNORM_NO_SHUFFLE"]}
def build():
    os.makedirs(OUT, exist_ok=True)
    for name, flags in VARIANTS.items():
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared",
                        "-Wno-unused-value", *flags, os.path.join(CSRC, "norm.hip"), os.path.join(CSRC, "error.hip"),
                        "-o", os.path.join(OUT, f"libnorm_{name}.so")], check=True)
def main():
    if not torch.cuda.is_available():
        build(); print("built"); return
    from paradis_model_amd import _lib
    B, C, P = 32, 1152, 2048
    x = torch.randn(B, C, P, device="cuda"); gy = torch.randn_like(x); w = torch.randn(C, device="cuda")
    mean = torch.randn(B, P, device="cuda"); rstd = torch.rand(B, P, device="cuda") + 0.5
    gx = torch.empty_like(x); gw = torch.empty(C, device="cuda"); gb = torch.empty(C, device="cuda")
    ws = torch.empty(64 << 20, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: ctypes.c_void_p(t.data_ptr())
    for name in VARIANTS:
        L = ctypes.CDLL(os.path.join(OUT, f"libnorm_{name}.so"))
        L.paradis_channel_norm_bwd.argtypes = _lib.SIGNATURES["paradis_channel_norm_bwd"][1]
        fn = lambda: L.paradis_channel_norm_bwd(p(gy), p(x), None, p(w), p(mean), p(rstd), p(gx), None, p(gw), p(gb),
                                                B, C, 0, P, C * P, 0, C * P, 0, p(ws), st)
        assert fn() == 0; torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name:12s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
if __name__ == "__main__":
    main()
