#!/usr/bin/env python3
"""Diagnostic: where the HOST time of one eager training step goes (cProfile of the enqueue of one step, GPU idle-queued):
    python tools/host_profile.py [--amp]       top functions by cumulative and by own time"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paradis_model_amd.config import default_config, feature_layout, stub_datamodule
from paradis_model_amd.harness import TrainStep, make_grids, synthetic_batch
from paradis_model_amd.loss import build_loss
from paradis_model_amd.model import Paradis

cfg = default_config()
lay = feature_layout(cfg)
lat_deg, lg, og = make_grids(32, 64, False)
torch.manual_seed(42)
model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
step = TrainStep(model, build_loss(cfg, lat_deg).cuda(), cfg, num_common=lay.num_common_features, n_inputs=cfg.dataset.n_time_inputs,
                 amp="--amp" in sys.argv)
batch = synthetic_batch(32, 64, False, 32, 1, seed=1, device="cuda")
for _ in range(3):
    step(batch)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(batch); ts.append(time.perf_counter() - t0)
print("host ms per eager step (enqueue only):", [round(1e3 * t, 1) for t in ts])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable(); step(batch); pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
