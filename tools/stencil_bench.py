import os
# needs the development build of the library (make -C paradis_model_amd/csrc dev): the shipped one exports
# no paradis_debug_set_* tunables
os.environ.setdefault("PARADIS_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                      "paradis_model_amd", "libparadis_hip_dev.so"))

#!/usr/bin/env python3
"""Diagnostic: depthwise geo-conv and ChannelNorm kernels in isolation (cfg2 shapes), HIP-event times
and effective bandwidth against the algorithmic bytes.  Usage on the GPU box: python tools/stencil_bench.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, C, H, W = 32, 1024, 32, 64
    x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    w = torch.randn(C, 1, 5, 5, device="cuda", requires_grad=True)
    gy = torch.randn(B, C, H, W, device="cuda")
    nb = x.numel() * 4
    from paradis_model_amd._lib import lib, dptr, stream_ptr
    y = torch.empty_like(x); gx = torch.empty_like(x); gw = torch.empty_like(w)
    st = stream_ptr()
    t = timeit(lambda: lib.paradis_dwconv_geo_fwd(dptr(x), dptr(w), None, dptr(y), B, C, H, W, 5, st))
    print(f"dwconv fwd   {t:7.1f} us  {2 * nb / t / 1e6:6.2f} TB/s")
    t = timeit(lambda: lib.paradis_dwconv_geo_dgrad(dptr(gy), dptr(w), dptr(gx), B, C, H, W, 5, st))
    print(f"dwconv dgrad {t:7.1f} us  {2 * nb / t / 1e6:6.2f} TB/s")
    ws = torch.empty(lib.paradis_dwconv_geo_wgrad_ws_bytes(B, C, H, W, 5) // 4 + 64, device="cuda")
    t = timeit(lambda: lib.paradis_dwconv_geo_wgrad(dptr(gy), dptr(x), dptr(gw), None, B, C, H, W, 5, dptr(ws), st))
    print(f"dwconv wgrad {t:7.1f} us  {2 * nb / t / 1e6:6.2f} TB/s")
    for Cn, extra in ((1024, 0), (1024, 128)):
        xx = torch.randn(B, Cn, H, W, device="cuda", requires_grad=True)
        xe = torch.randn(B, extra, H, W, device="cuda", requires_grad=True) if extra else None
        wn = torch.randn(Cn + extra, device="cuda", requires_grad=True)
        bn = torch.randn(Cn + extra, device="cuda", requires_grad=True)
        g = torch.randn(B, Cn + extra, H, W, device="cuda")
        nbn = (Cn + extra) * B * H * W * 4
        yy = None

        def fwd():
            nonlocal yy
            yy = ops.channel_norm(xx, wn, bn, 1e-5, xe)
        for px in (64, 32):
            lib.paradis_debug_set_norm_fwd_px(px)
            t = timeit(fwd)
            print(f"channel_norm fwd C={Cn}+{extra} px={px} {t:7.1f} us  {2 * nbn / t / 1e6:6.2f} TB/s")

        def bwd():
            xx.grad = None; wn.grad = None; bn.grad = None
            if xe is not None:
                xe.grad = None
            yy.backward(g, retain_graph=True)
        for rr in (0, 1):
            lib.paradis_debug_set_norm_bwd_reread(rr)
            t = timeit(bwd)
            print(f"channel_norm bwd C={Cn}+{extra} reread={rr} {t:7.1f} us  {3 * nbn / t / 1e6:6.2f} TB/s (incl. autograd glue)")


if __name__ == "__main__":
    main()
