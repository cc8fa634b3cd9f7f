#!/usr/bin/env python3
"""Diagnostic: host time to enqueue one training step (no synchronisation inside) vs its GPU time:
how far the Python/ctypes launch path is from becoming the bottleneck (e.g. under DDP hooks)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd.config import default_config, feature_layout, stub_datamodule
from paradis_model_amd.harness import TrainStep, make_grids, synthetic_batch
from paradis_model_amd.loss import build_loss
from paradis_model_amd.model import Paradis
cfg = default_config(); lay = feature_layout(cfg)
lat_deg, lg, og = make_grids(32, 64, False)
torch.manual_seed(0)
model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
step = TrainStep(model, build_loss(cfg, lat_deg).cuda(), cfg, num_common=lay.num_common_features, n_inputs=2)
for B in (32, 4):
    batch = synthetic_batch(32, 64, False, B, 1, device="cuda")
    for _ in range(3):
        step(batch)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        step(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print(f"B={B}: host enqueue {1e3 * min(host):.1f} ms, step {1e3 * min(total):.1f} ms")
