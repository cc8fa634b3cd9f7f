"""Diagnostic: the depthwise 5x5 geo-convolution kernels (forward, data gradient, weight gradient, both gradients in one
call with and without the addend) in isolation at the layer shapes of the default model - 32x64 B = 32, 128x256 B = 8
(C = 1024 / 384) and 721x1440 B = 1 -, HIP-event times against the algorithmic bytes, and a parity check of the forward
against the oracle's padded conv2d.  PARADIS_HIP_LIB selects the build (tools/build_variant.sh ... "-DDWCONV_TILES=0" =
the one-tile-per-workgroup kernels on the larger grids)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd._lib import lib, dptr, stream_ptr
from oracle import paradis_oracle as O

name = os.path.basename(os.environ.get("PARADIS_HIP_LIB", "shipped"))


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, C, H, W) in ((32, 1024, 32, 64), (32, 384, 32, 64), (3, 10, 29, 64), (8, 1024, 128, 256), (8, 384, 128, 256),
                     (1, 1024, 721, 1440)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, C, H, W, device="cuda", generator=g)
    w = torch.randn(C, 1, 5, 5, device="cuda", generator=g)
    gy = torch.randn(B, C, H, W, device="cuda", generator=g)
    y, gx, gw = torch.empty_like(x), torch.empty_like(x), torch.empty_like(w)
    ad = torch.randn(B, C, H, W, device="cuda", generator=g)
    st = stream_ptr()
    nb = x.numel() * 4
    ws = torch.empty(lib.paradis_dwconv_geo_wgrad_ws_bytes(B, C, H, W, 5) // 4 + 64, device="cuda")
    big = torch.randn(64 << 20, device="cuda")
    for _ in range(100):
        big = big * 1.0001
    tf = timeit(lambda: lib.paradis_dwconv_geo_fwd(dptr(x), dptr(w), None, dptr(y), B, C, H, W, 5, st))
    td = timeit(lambda: lib.paradis_dwconv_geo_dgrad(dptr(gy), dptr(w), dptr(gx), B, C, H, W, 5, st))
    tw = timeit(lambda: lib.paradis_dwconv_geo_wgrad(dptr(gy), dptr(x), dptr(gw), None, B, C, H, W, 5, dptr(ws), st))
    tb = timeit(lambda: lib.paradis_dwconv_geo_bwd(dptr(gy), dptr(x), dptr(w), None, dptr(gx), dptr(gw), None, B, C, H, W, 5,
                                                   dptr(ws), st))
    ta = timeit(lambda: lib.paradis_dwconv_geo_bwd(dptr(gy), dptr(x), dptr(w), dptr(ad), dptr(gx), dptr(gw), None, B, C, H, W,
                                                   5, dptr(ws), st))
    # parity: forward, and the two gradients through autograd of the same padded convolution (fp64)
    cs = min(C, 8)
    xs, ws_ = x[:2, :cs].double().cpu(), w[:cs].double().cpu()
    ref = torch.nn.functional.conv2d(O.geocyclic_pad(xs, 2), ws_, groups=cs)
    lib.paradis_dwconv_geo_fwd(dptr(x), dptr(w), None, dptr(y), B, C, H, W, 5, st)
    ef = float((y[:2, :cs].double().cpu() - ref).abs().max() / ref.abs().max())
    print("%-14s B=%d C=%d %dx%d: fwd %.1f us (%.2f TB/s)  dgrad %.1f us (%.2f)  wgrad %.1f us (%.2f)  both %.1f us (%.2f)"
          "  both+addend %.1f us (%.2f)  fwd err %.1e"
          % (name, B, C, H, W, tf, 2 * nb / tf / 1e6, td, 2 * nb / td / 1e6, tw, 2 * nb / tw / 1e6, tb, 3 * nb / tb / 1e6,
             ta, 4 * nb / ta / 1e6, ef), flush=True)
