#!/usr/bin/env python3
"""Diagnostic: every parameter gradient of the default model (32x64, B=2) on the HIP path against the
fp64 oracle, next to the CPU-fp32 oracle's own distance to fp64 (max-rel and rms-rel, worst ratios)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_hip_model import _build, _oracle_grads
from oracle import paradis_oracle as O
from paradis_model_amd.config import default_config, feature_layout
from tests._util import make_grid, max_rel, seeded
cfg = default_config()
_, lg, og = make_grid(32, 64, False)
model = _build(cfg, lg, og)
with torch.no_grad():
    g = torch.Generator().manual_seed(7)
    for n, p in model.named_parameters():
        if n.endswith((".A", ".U", ".V")):
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
lay = feature_layout(cfg)
spec = O.spec_from_cfg(cfg, 32, 64, lay.num_in_dyn_features, lay.num_in_static_features, lay.num_out_features)
x = seeded(5, 2, 186, 32, 64); x[:, -2], x[:, -1] = lg, og
ct = seeded(6, 2, lay.num_out_features, 32, 64)
y32, g32 = _oracle_grads(model, spec, x, ct, lg, og, torch.float32)
_, g64 = _oracle_grads(model, spec, x, ct, lg, og, torch.float64)
got = model(x.cuda()); (got * ct.cuda()).sum().backward()
rows = []
for n, p in model.named_parameters():
    ref = g64.get(n)
    if ref is None or float(ref.abs().max()) == 0: continue
    eg = max_rel(p.grad.cpu().double(), ref); ec = max_rel(g32[n].double(), ref)
    # rms-relative too
    rg = float(((p.grad.cpu().double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
    rc = float(((g32[n].double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
    rows.append((eg / max(ec, 1e-12), n, eg, ec, rg, rc))
rows.sort(reverse=True)
print("top ratios (max-rel gpu/cpu):")
for r in rows[:12]: print("  %5.2f %-50s max gpu %.2e cpu %.2e | rms gpu %.2e cpu %.2e" % r)
import statistics
print("median ratio", statistics.median(r[0] for r in rows), "params", len(rows))
print("rms ratio top:", sorted(((r[4] / max(r[5], 1e-12)), r[1]) for r in rows)[-5:])
