"""Diagnostic: ChannelNorm forward in isolation at the layer shapes of the default model (C = 1024, the reaction
block's 1024 + 128 virtual concat, the velocity net's 384), with a parity check against torch.  PARADIS_HIP_LIB selects
the build (tools/build_variant.sh ... norm.hip "-DNORM_FWD32=0")."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops

name = os.path.basename(os.environ.get("PARADIS_HIP_LIB", "shipped"))
for (B, H, W) in ((32, 32, 64), (8, 128, 256), (2, 33, 50)):
    for C1, C2 in ((1024, 0), (1024, 128), (384, 0)):
        x1 = torch.randn(B, C1, H, W, device="cuda") * 3 + 0.5
        x2 = torch.randn(B, C2, H, W, device="cuda") if C2 else None
        C = C1 + C2
        w, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        y, mean, rstd = ops._channel_norm(x1, x2, w, b, 1e-5)
        xc = (torch.cat([x1, x2], 1) if C2 else x1).double()
        var, mu = torch.var_mean(xc, dim=1, keepdim=True)
        ref = (xc - mu) * (var + 1e-5).rsqrt() * w.double()[None, :, None, None] + b.double()[None, :, None, None]
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        big = torch.randn(64 << 20, device="cuda")
        for _ in range(200):
            big = big * 1.0001
        for _ in range(5):
            ops._channel_norm(x1, x2, w, b, 1e-5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops._channel_norm(x1, x2, w, b, 1e-5)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 50 * 1e3
        alg = 8.0 * B * H * W * C
        print("%-14s B=%d %dx%d C=%d+%d: %.1f us  (%.2f TB/s of algorithmic bytes)  max err vs fp64 %.1e"
              % (name, B, H, W, C1, C2, t, alg / t / 1e6, err), flush=True)
