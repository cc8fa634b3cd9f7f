"""Diagnostic: ChannelNorm backward in isolation at the layer shape of the default model (C = 1024 and the reaction
block's 1024 + 128 virtual concat), with and without the fused residual addend.  PARADIS_HIP_LIB selects the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops

name = os.path.basename(os.environ.get("PARADIS_HIP_LIB", "shipped"))
for (B, H, W) in ((32, 32, 64), (8, 128, 256)):
    for C1, C2, add in ((1024, 0, True), (1024, 128, True), (1024, 0, False)):
        x1 = torch.randn(B, C1, H, W, device="cuda")
        x2 = torch.randn(B, C2, H, W, device="cuda") if C2 else None
        C = C1 + C2
        w, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        y, mean, rstd = ops._channel_norm(x1, x2, w, b, 1e-5)
        gy = torch.randn_like(y)
        addend = torch.randn_like(x1) if add else None
        big = torch.randn(64 << 20, device="cuda")
        for _ in range(300):
            big = big * 1.0001
        for _ in range(5):
            ops._channel_norm_backward(gy, x1, x2, w, mean, rstd, addend)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops._channel_norm_backward(gy, x1, x2, w, mean, rstd, addend)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 30 * 1e3
        alg = 4.0 * B * H * W * (2 * C + C + (C1 if add else 0))     # gy, x read once; gx written; addend read
        print("%-14s B=%d %dx%d C=%d+%d add=%d: %.1f us  (%.2f TB/s of algorithmic bytes)" % (name, B, H, W, C1, C2, add, t, alg / t / 1e6), flush=True)
