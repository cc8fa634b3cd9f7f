import sys, os, torch
sys.path.insert(0, '/root/repo')
from tests.test_hip_graph import _setup
from paradis_model_amd.harness import GraphedTrainStep
model, step, batches = _setup(True)
g = GraphedTrainStep(step, batches[0], warmup=2)
named = dict(model.named_parameters())
def bad():
    return [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in named.items() if p.grad is not None and not torch.isfinite(p.grad).all()]
for i in range(3):
    g(batches[i % 2]); torch.cuda.synchronize()
print("A", bad())
mode = os.environ.get("DBG_MODE", "randn")
if mode == "randn":
    junk = [torch.randn(1 << 20, device="cuda") for _ in range(8)]
elif mode == "empty":
    junk = [torch.empty(1 << 20, device="cuda") for _ in range(8)]
elif mode == "small":
    junk = [torch.empty(16, device="cuda") for _ in range(8)]
elif mode == "sync":
    torch.cuda.synchronize()
g(batches[0]); torch.cuda.synchronize(); print("B", mode, bad()[:6])
