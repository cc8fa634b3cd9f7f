"""Diagnostic: the GlobalBias adjoint chain in isolation - projection adjoint (gPw, gm8 from the bias-map gradient) and
the rank-R tail (gA, gU, gV from gm8) - at the three reference grids, Co = 1024, Cin = 8, R = 128.  PARADIS_HIP_LIB selects
the build (tools/build_variant.sh ... misc.hip "-DGBIAS_GPW_ROWS=0" = one workgroup per (o, c) on every grid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops

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


for (H, W) in ((32, 64), (128, 256), (721, 1440)):
    for Co in (1024, 384):
        Cin, R = 8, 128
        g = torch.Generator(device="cuda").manual_seed(1)
        r = lambda *s: torch.randn(*s, device="cuda", generator=g)
        gmap, m8, pw = r(Co, H, W), r(Cin, H, W), r(Co, Cin)
        A, U, V, gm8 = r(Cin, R), r(R, H), r(R, W), r(Cin, H, W)
        tp = timeit(lambda: ops._global_bias_proj_backward(gmap, m8, pw))
        tt = timeit(lambda: ops._global_bias_m8_backward(gm8, A, U, V))
        gpw, _ = ops._global_bias_proj_backward(gmap, m8, pw)
        ref = torch.einsum("ohw,chw->oc", gmap.double(), m8.double())
        e1 = float((gpw.double() - ref).abs().max() / ref.abs().max())
        gA, gU, gV = ops._global_bias_m8_backward(gm8, A, U, V)
        rA = torch.einsum("chw,rh,rw->cr", gm8.double(), U.double(), V.double())
        rU = torch.einsum("chw,cr,rw->rh", gm8.double(), A.double(), V.double())
        rV = torch.einsum("chw,cr,rh->rw", gm8.double(), A.double(), U.double())
        e2 = max(float((a.double() - b).abs().max() / b.abs().max()) for a, b in ((gA, rA), (gU, rU), (gV, rV)))
        print("%-14s %dx%d Co=%d: projection adjoint %.1f us, rank tail %.1f us (host-bound below ~40 us); max err vs fp64 %.1e / %.1e"
              % (name, H, W, Co, tp, tt, e1, e2), flush=True)
