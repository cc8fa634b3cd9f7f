"""Diagnostic: accuracy of the pointwise GEMM (forward, data gradient, weight gradient) against an fp64 evaluation
at the default model's layer shapes, for the library named by PARADIS_HIP_LIB and the arithmetic of PARADIS_GEMM.
    python tools/gemm_wide_check.py [H W B]
One line per shape: max |err| / max |ref| of y, dX, dW."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd import ops

H, W, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 256, 1)
SHAPES = [(1024, 186), (384, 1024), (1536, 384), (768, 1024), (1024, 768), (1024, 1024), (896, 1152), (896, 896),
          (1024, 896), (97, 1024), (128, 64)]


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def main():
    tag = "%s gemm=%s" % (os.path.basename(os.environ.get("PARADIS_HIP_LIB", "shipped")), os.environ.get("PARADIS_GEMM", "default"))
    g = torch.Generator().manual_seed(3)
    worst = [0.0, 0.0, 0.0]
    for co, ci in SHAPES:
        # operands with a wide dynamic range per element, as activations behind a ChannelNorm / SiLU have
        x = (torch.randn(B, ci, H, W, generator=g) * torch.exp(2.0 * torch.randn(B, ci, 1, 1, generator=g))).cuda()
        w = (torch.randn(co, ci, generator=g) / ci ** 0.5).cuda()
        b = torch.randn(co, generator=g).cuda()
        ct = torch.randn(B, co, H, W, generator=g).cuda()
        x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
        y = ops.pointwise(x, w, b)
        y.backward(ct)
        xd, wd = x.detach().double(), w.detach().double()
        yr = torch.einsum("oc,bchw->bohw", wd, xd) + b.detach().double()[None, :, None, None]
        dxr = torch.einsum("oc,bohw->bchw", wd, ct.double())
        dwr = torch.einsum("bohw,bchw->oc", ct.double(), xd)
        e = (rel(y.detach(), yr), rel(x.grad, dxr), rel(w.grad, dwr))
        worst = [max(a, c) for a, c in zip(worst, e)]
        print("[%s] %4dx%-4d y %.2e dX %.2e dW %.2e" % (tag, co, ci, *e))
    print("[%s] worst y %.2e dX %.2e dW %.2e" % (tag, *worst))


if __name__ == "__main__":
    main()
