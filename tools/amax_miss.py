"""Diagnostic: which GEMM operands still take their own amax read pass (f16x2 scheme) in one training step of the
default model - count, (blocks, floats per block) and the Python call chain of every paradis_amax_partials launch
that is not the weights'.  Round 2: 3 per step (the raw model input and the loss gradient); 43 before the backward kernels of the advection,
ChannelNorm and the gated blend got their side outputs, 311 without any."""
import collections, sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paradis_model_amd import ops, _lib
from paradis_model_amd.config import default_config, stub_datamodule
from paradis_model_amd.harness import TrainStep, make_grids, synthetic_batch
from paradis_model_amd.loss import build_loss
from paradis_model_amd.model import Paradis
miss = collections.Counter()
orig = _lib.lib.paradis_amax_partials
def spy(*a):
    st = traceback.extract_stack(limit=9)
    key = " <- ".join(f"{f.name}:{f.lineno}" for f in st[-2:-7:-1])
    miss[(a[1], a[2], key)] += 1
    return orig(*a)
_lib.lib.paradis_amax_partials = spy
ops.lib.paradis_amax_partials = spy
cfg = default_config()
lat, lg, og = make_grids(32, 64, False)
torch.manual_seed(0)
model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
loss = build_loss(cfg, lat).cuda()
step = TrainStep(model, loss, cfg)
batch = synthetic_batch(32, 64, False, 4, 1, device="cuda")
step(batch); miss.clear(); step(batch)
for k, v in sorted(miss.items(), key=lambda kv: -kv[1]):
    print(v, k)
