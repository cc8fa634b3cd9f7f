"""Diagnostic: is the error of the pointwise GEMM arithmetics CORRELATED (a systematic offset) or zero-mean?
For y, dX and dW at a few layer shapes: the SIGNED error against fp64 in units of u = 2^-24 * rms(reference):
mean (an offset every output shares), rms, and the mean of err * checkerboard sign.
A zero-mean error of rms r averages down as r / sqrt(n) in the sums that follow (weight gradients over 32,768
points, ChannelNorm statistics); an offset does not.
    python tools/gemm_bias_check.py [H W B]      (PARADIS_HIP_LIB / schemes are looped over inside)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops  # noqa: E402

H, W, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 256, 1)
SHAPES = [(1024, 1024), (896, 1152), (384, 1024), (1024, 768)]


def stats(got, ref, checker=True):
    """(mean, rms, mean of err * checkerboard sign) in units of u; the checkerboard is the sign pattern of the
    library's negated-space blocks: (-1)^(row / 32 + column / 64) over (channel, pixel) of a [B, C, H, W] output"""
    err = got.double() - ref
    u = float(ref.pow(2).mean().sqrt()) * 2.0 ** -24
    chk = float("nan")
    if checker and err.dim() == 4:
        Bn, C, Hh, Ww = err.shape
        rs = 1.0 - 2.0 * ((torch.arange(C, device=err.device) >> 5) & 1).double()
        cs = 1.0 - 2.0 * ((torch.arange(Hh * Ww, device=err.device) >> 6) & 1).double()
        chk = float((err.reshape(Bn, C, Hh * Ww) * rs[None, :, None] * cs[None, None, :]).mean()) / u
    return float(err.mean()) / u, float(err.pow(2).mean().sqrt()) / u, chk


def main():
    g = torch.Generator().manual_seed(3)
    for positive in (False, True):
        for co, ci in SHAPES:
            x = torch.randn(B, ci, H, W, generator=g)
            w = torch.randn(co, ci, generator=g) / ci ** 0.5
            ct = torch.randn(B, co, H, W, generator=g)
            if positive:    # everything positive: growing sums, where an alignment bias cannot hide behind cancellation
                x, w, ct = x.abs(), w.abs(), ct.abs()
            x, w, ct = x.cuda(), w.cuda(), ct.cuda()
            xd, wd = x.double(), w.double()
            yr = torch.einsum("oc,bchw->bohw", wd, xd)
            dxr = torch.einsum("oc,bohw->bchw", wd, ct.double())
            dwr = torch.einsum("bohw,bchw->oc", ct.double(), xd)
            for name, scheme in (("exact", ops.GEMM_EXACT), ("bf16x3", ops.GEMM_BF16X3)):
                xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
                y = ops.pointwise(xx, ww, None, scheme=scheme)
                y.backward(ct)
                s = [stats(y.detach(), yr), stats(xx.grad, dxr), stats(ww.grad, dwr)]
                print("%s %4dx%-4d %-6s  y mean %+7.3f rms %6.2f chk %+7.3f | dX mean %+7.3f rms %6.2f chk %+7.3f | "
                      "dW mean %+8.3f rms %7.2f chk %+8.3f   [u = 2^-24 rms(ref)]"
                      % ("pos " if positive else "rand", co, ci, name, *s[0], *s[1], *s[2]))


if __name__ == "__main__":
    main()
