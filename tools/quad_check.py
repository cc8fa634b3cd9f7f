#!/usr/bin/env python3
"""PARADIS_GEMM_QUAD=1 (256 x 256 bf16x3 tile, activation tile staged once for two m-tiles) against the 128 x 256 kernel:
bitwise equality of forward / dgrad outputs over the layer shapes (run twice, once per setting, compare saved results)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops
out = sys.argv[1]
g = torch.Generator().manual_seed(3)
res = {}
for (B, Ci, Co, H, W) in ((2, 1024, 1024, 32, 64), (2, 1152, 896, 32, 64), (1, 896, 896, 32, 64), (2, 384, 1536, 32, 64),
                          (2, 1024, 384, 32, 64), (1, 186, 1024, 32, 64), (1, 1024, 97, 32, 64), (1, 768, 1024, 128, 256),
                          (3, 130, 258, 9, 20)):
    x = torch.randn(B, Ci, H, W, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(Co, generator=g).cuda()
    ct = torch.randn(B, Co, H, W, generator=g).cuda()
    y = ops.pointwise(x, w, b, act="SiLU", scheme=ops.GEMM_BF16X3)
    y.backward(ct)
    res[(B, Ci, Co, H, W)] = (y.detach().cpu(), x.grad.cpu(), w.grad.cpu())
torch.save(res, out)
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for k in res:
        eq = [bool(torch.equal(a, b)) for a, b in zip(res[k], ref[k])]
        print(k, "bitwise equal (y, gx, gw):", eq)
