"""GPU: the opt-in side stream of the weight-gradient GEMMs (``ops.WgradSide``, PARADIS_WGRAD_STREAM=1; round 6, verdict r5
item 2).  Measured, it buys nothing - the chip is work-conserving under co-scheduling (profiles/r06_overlap*) - so it
ships switched off; what is pinned here is that switching it on changes no bit and breaks no ordering: eager, under a
captured HIP graph, and with a second backward pass reusing the stream."""
import pytest
import torch

from paradis_model_amd.config import reduced_config, stub_datamodule
from tests._util import make_grid

pytestmark = pytest.mark.gpu


def _setup(capturable=False):
    from paradis_model_amd.harness import TrainStep
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis
    cfg = reduced_config()
    lat_deg, lg, og = make_grid(16, 32, False)
    torch.manual_seed(42)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
    return model, TrainStep(model, build_loss(cfg, lat_deg).cuda(), cfg, capturable=capturable)


def _params(m):
    return torch.cat([p.detach().flatten() for p in m.parameters()]).clone()


def test_side_stream_weight_gradients_change_no_bit():
    from paradis_model_amd import ops
    from paradis_model_amd.harness import GraphedTrainStep, synthetic_batch
    batch = synthetic_batch(16, 32, False, 2, 1, seed=5, device="cuda")
    keep = ops.WgradSide.enabled
    try:
        out = {}
        for side in (False, True):
            ops.WgradSide.enabled = side
            m, step = _setup()
            losses = [float(step(batch)) for _ in range(4)]
            torch.cuda.synchronize()
            assert not ops.WgradSide._pending          # joined at the end of every backward pass
            out[side] = (_params(m), losses)
        assert out[True][1] == out[False][1] and torch.equal(out[True][0], out[False][0])
        # captured: the fork to the side stream and the join become edges of the graph
        ops.WgradSide.enabled = True
        mg, sg = _setup(capturable=True)
        g = GraphedTrainStep(sg, batch, warmup=2)
        for _ in range(2):
            g(batch)
        torch.cuda.synchronize()
        assert torch.equal(_params(mg), out[False][0])      # 2 eager warm-up steps + 2 replays = the 4 eager steps
    finally:
        ops.WgradSide.enabled = keep
        ops.WgradSide.join()
