#!/usr/bin/env python3
"""Diagnostic: semi-Lagrangian displacement statistics INSIDE the default model (random init, synthetic N(0,1) input -
the bench's workload) at the large grids, in padded-grid cells: what a window halo of the tile-row advection schedule
has to cover.  Per layer: quantiles of |dx|, |dy| (departure - arrival, cells), share of points whose tap block leaves
a halo of 6 / 8 / 10 / 16 / 24 cells, the same for polar rows (|lat| > 75 deg) only, and how much of the displacement
is the tile mean (64 x 128 tiles): mean |tile-mean dx| against the mean |dx - tile mean|.

  python tools/adv_disp_stats.py 128 256 [B]     |     python tools/adv_disp_stats.py 721 1440
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd import ops  # noqa: E402
from paradis_model_amd.config import default_config, stub_datamodule  # noqa: E402
from paradis_model_amd.harness import make_grids, synthetic_batch, assemble_model_input  # noqa: E402
from paradis_model_amd.model import Paradis  # noqa: E402

H, W = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
poles = H % 2 == 1
cfg = default_config()
_, lg, og = make_grids(H, W, poles)
torch.manual_seed(42)
model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
batch = synthetic_batch(H, W, poles, B, 1, device="cuda")
inp, tgt, forc, const = batch
mi = assemble_model_input(inp, forc.permute(0, 1, 4, 2, 3)[:, 0].unsqueeze(1), const[:, :1].permute(0, 1, 4, 2, 3))

orig = ops.sl_advect_vel
layer = [0]


def q(t, qs):
    t = t.flatten().float()
    if t.numel() > 4_000_000:
        t = t[torch.randint(0, t.numel(), (4_000_000,), device=t.device)]
    return [float(x) for x in torch.quantile(t, torch.tensor(qs, device=t.device))]


def hook(field, vel, geom, dt, mode="bicubic", flags=None):
    K = field.shape[1]
    u, v = vel[:, :K].double(), vel[:, K:].double()
    lat = lg.cuda().double()[None, None]
    lam, phi = -u * dt, -v * dt
    sa, ca = torch.sin(lat), torch.cos(lat)
    s = torch.sin(phi) * ca + torch.cos(phi) * torch.cos(lam) * sa
    latd = torch.asin(s.clamp(-1 + 1e-7, 1 - 1e-7))
    n = torch.cos(phi) * torch.sin(lam)
    d = torch.cos(phi) * torch.cos(lam) * ca - torch.sin(phi) * sa
    dlon = torch.atan2(n, d)
    cx = (W - 1) / geom.d_lon
    cy = (H - 1) / geom.d_lat
    dx = (dlon * cx)
    dy = ((latd - lat) * cy)
    adx, ady = dx.abs(), dy.abs()
    qs = [0.5, 0.9, 0.99, 0.999]
    polar = (lat.abs() > math.radians(75.0)).expand_as(adx)
    print(f"layer {layer[0]}: |u dt| rms {float((u * dt).pow(2).mean().sqrt()):.4f} rad; |dx| q50/90/99/99.9 "
          + "/".join(f"{x:.1f}" for x in q(adx, qs)) + f" max {float(adx.max()):.0f};  |dy| "
          + "/".join(f"{x:.1f}" for x in q(ady, qs)) + f" max {float(ady.max()):.0f}")
    m = torch.maximum(adx, ady)
    outs = {h: float((m > h).double().mean()) for h in (6, 8, 10, 16, 24, 32)}
    outs_np = {h: float(((m > h) & ~polar).double().sum() / (~polar).double().sum()) for h in (6, 8, 10, 16, 24, 32)}
    outs_y = {h: float((ady > h).double().mean()) for h in (6, 8, 10, 16, 24)}
    print("   share outside halo (all rows): " + " ".join(f"{h}:{x:.4f}" for h, x in outs.items()))
    print("   share outside halo (|lat|<75): " + " ".join(f"{h}:{x:.4f}" for h, x in outs_np.items()))
    print("   share with |dy| outside      : " + " ".join(f"{h}:{x:.4f}" for h, x in outs_y.items()))
    # tile-mean vs residual (64 x 128 tiles)
    th, tw = 64, 128
    Hc, Wc = (H // th) * th, (W // tw) * tw
    if Hc and Wc:
        t = dx[:, :, :Hc, :Wc].reshape(dx.shape[0], K, Hc // th, th, Wc // tw, tw)
        tm = t.mean(dim=(3, 5), keepdim=True)
        ty = dy[:, :, :Hc, :Wc].reshape(dx.shape[0], K, Hc // th, th, Wc // tw, tw)
        tmy = ty.mean(dim=(3, 5), keepdim=True)
        print(f"   tiles 64x128: mean |tile-mean dx| {float(tm.abs().mean()):.2f}, mean |dx - tile mean| {float((t - tm).abs().mean()):.2f};"
              f" dy: {float(tmy.abs().mean()):.2f} / {float((ty - tmy).abs().mean()):.2f}")
    layer[0] += 1
    return orig(field, vel, geom, dt, mode, flags)


ops.sl_advect_vel = hook
with torch.no_grad():
    y = model(mi)
torch.cuda.synchronize()
print("output finite:", bool(torch.isfinite(y).all()))
