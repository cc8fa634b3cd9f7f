"""PARADIS_DETERMINISTIC=1: two backward passes of the same step are bit-identical (the reference's CPU
path is deterministic; by default a few of the GPU reductions finish with float atomics).  The variable
is read when the library is first used, hence the child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import torch
from paradis_model_amd.config import default_config, reduced_config, stub_datamodule
from paradis_model_amd.harness import make_grids, synthetic_batch, rollout_loss
from paradis_model_amd.loss import build_loss
from paradis_model_amd.model import Paradis

def grads(cfg, B, S):
    lat_deg, lg, og = make_grids(32, 64, False)
    torch.manual_seed(42)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        for n, p in model.named_parameters():
            if n.endswith((".A", ".U", ".V")):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    loss_fn = build_loss(cfg, lat_deg).cuda()
    batch = synthetic_batch(32, 64, False, B, S, device="cuda")
    out = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        total, _ = rollout_loss(model, loss_fn, batch, num_common=83, n_inputs=2)
        out.append((total.clone(), [p.grad.clone() for p in model.parameters()]))
    return model, out

for name, cfg, B, S in (("reduced", reduced_config(), 3, 2), ("default", default_config(), 2, 1)):
    model, (a, b) = grads(cfg, B, S)
    same = torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    worst = max(float((x - y).abs().max()) for x, y in zip(a[1], b[1]))
    print(name, "IDENTICAL" if same else "DIFFERENT", worst)
"""


def _run(env_value):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()      # the child needs HBM the parent's caching allocator may be holding
    env = dict(os.environ, PARADIS_DETERMINISTIC=env_value)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return [line.split() for line in r.stdout.strip().splitlines() if line and line.split()[0] in ("reduced", "default")]


def test_two_backward_passes_are_bit_identical_in_deterministic_mode():
    rows = _run("1")
    print(rows)
    assert len(rows) == 2 and all(r[1] == "IDENTICAL" for r in rows), rows
