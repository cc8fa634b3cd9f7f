#!/usr/bin/env python3
"""Diagnostic: where do the small at::fill_/zero_ kernels of a training step come from (torch profiler)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paradis_model_amd.config import default_config, feature_layout, stub_datamodule
from paradis_model_amd.harness import TrainStep, make_grids, synthetic_batch
from paradis_model_amd.loss import build_loss
from paradis_model_amd.model import Paradis
cfg = default_config()
lay = feature_layout(cfg)
lat_deg, lg, og = make_grids(32, 64, False)
torch.manual_seed(0)
model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
step = TrainStep(model, build_loss(cfg, lat_deg).cuda(), cfg, num_common=lay.num_common_features, n_inputs=2)
batch = synthetic_batch(32, 64, False, 4, 1, device="cuda")
for _ in range(2):
    step(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(batch)
torch.cuda.synchronize()
NAMES = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::copy_",
         "aten::clone", "aten::contiguous")
rows = [e for e in prof.events() if e.name in NAMES]
from collections import Counter
cnt = Counter()
for e in rows:
    st = [s for s in (e.stack or []) if "paradis_model_amd" in s or "torch/autograd" in s][:2]
    cnt[(e.name, str(e.input_shapes)[:60], tuple(st))] += 1
for (name, shp, st), n in cnt.most_common(40):
    print(n, name, shp, " | ".join(s.split("/")[-1] for s in st))
