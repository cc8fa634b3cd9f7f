"""GPU: the training step captured in a HIP graph (harness.GraphedTrainStep) replays to the same parameters and
losses as the eager step - forward, ParadisLoss, backward and the AdamW update inside ONE graph launch per step
(host time per step; reference trainer.py:498-587 is the step being captured)."""
import pytest
import torch

from paradis_model_amd.config import reduced_config, stub_datamodule
from tests._util import make_grid, max_rel

pytestmark = pytest.mark.gpu


def _setup(capturable):
    from paradis_model_amd.harness import TrainStep, synthetic_batch
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import Paradis
    cfg = reduced_config()
    lat_deg, lg, og = make_grid(16, 32, False)
    torch.manual_seed(42)
    model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
    step = TrainStep(model, build_loss(cfg, lat_deg).cuda(), cfg, capturable=capturable)
    batches = [synthetic_batch(16, 32, False, 2, 2, seed=5 + i, device="cuda") for i in range(2)]   # S = 2 rollout
    return model, step, batches


@pytest.mark.parametrize("gemm", ["bf16x3", "exact"])
def test_graphed_train_step_equals_eager(gemm, monkeypatch):
    from paradis_model_amd import ops
    from paradis_model_amd.harness import GraphedTrainStep
    monkeypatch.setattr(ops, "GEMM_SCHEME", ops._SCHEMES[gemm])
    n_steps, warm = 6, 2
    # eager reference: n_steps steps over alternating batches (the first `warm` on batch 0 like the warm-up below)
    model_e, step_e, batches = _setup(False)
    order = [0] * warm + [i % 2 for i in range(n_steps - warm)]
    losses_e = [float(step_e(batches[i])) for i in order]
    # graphed: warm-up steps (eager, on batch 0) + replays
    model_g, step_g, _ = _setup(True)
    g = GraphedTrainStep(step_g, batches[0], warmup=warm)
    losses_g = [float(g(batches[i])) for i in order[warm:]]
    torch.cuda.synchronize()
    for a, b in zip(losses_e[warm:], losses_g):
        assert abs(a - b) <= 1e-6 * abs(a), (losses_e, losses_g)
    pe = torch.cat([p.detach().flatten() for p in model_e.parameters()])
    pg = torch.cat([p.detach().flatten() for p in model_g.parameters()])
    assert max_rel(pg, pe) <= 1e-6
    # host-side optimiser state followed the replays
    st = step_g.opt.state[next(iter(model_g.parameters()))]
    assert int(st["step"]) == n_steps
    # a learning-rate change between replays reaches the captured update
    for grp in step_g.opt.param_groups:
        grp["lr"] = 0.0
    step_g.opt.sync_device_state()
    before = pg.clone()
    wd = step_g.opt.param_groups[0]["weight_decay"]
    g(batches[0])
    torch.cuda.synchronize()
    after = torch.cat([p.detach().flatten() for p in model_g.parameters()])
    assert torch.equal(after, before) or wd == 0 or max_rel(after, before) < 1e-12   # lr = 0: no update at all


def test_graphed_step_host_time_is_one_launch():
    """what the graph buys: the host returns from a replay long before an eager step has been enqueued"""
    import time
    from paradis_model_amd.harness import GraphedTrainStep
    model_e, step_e, batches = _setup(False)
    for _ in range(3):
        step_e(batches[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step_e(batches[0])
    t_eager = (time.perf_counter() - t0) / 5          # host time to enqueue (no sync inside)
    torch.cuda.synchronize()
    model_g, step_g, _ = _setup(True)
    g = GraphedTrainStep(step_g, batches[0], warmup=2)
    g(batches[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g(batches[0])
    t_graph = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    print("host time per step: eager %.2f ms, graph replay %.2f ms" % (1e3 * t_eager, 1e3 * t_graph))
    assert t_graph < 0.5 * t_eager


def test_eager_forward_between_replays_sees_current_weights():
    """A validation forward between two replays must run on the weights the replay just wrote: the replay updates the
    parameters through raw pointers (no autograd version bump, no optimiser post-hook), so the cached bf16x3 weight
    images have to be invalidated by GraphedTrainStep itself.  Twin: the same steps and forwards eagerly."""
    from paradis_model_amd.harness import GraphedTrainStep, assemble_model_input
    model_e, step_e, batches = _setup(False)
    model_g, step_g, _ = _setup(True)

    def fwd(model, batch):
        inp, tgt, forc, const = batch
        mi = assemble_model_input(inp, forc.permute(0, 1, 4, 2, 3)[:, 0].unsqueeze(1), const[:, :1].permute(0, 1, 4, 2, 3))
        with torch.no_grad():
            return model(mi)

    warm = 2
    g = GraphedTrainStep(step_g, batches[0], warmup=warm)
    for _ in range(warm):
        step_e(batches[0])
    outs_e, outs_g = [], []
    for i in range(3):
        step_e(batches[i % 2]); outs_e.append(fwd(model_e, batches[1]))
        g(batches[i % 2]); outs_g.append(fwd(model_g, batches[1]))
    torch.cuda.synchronize()
    for a, b in zip(outs_e, outs_g):
        assert max_rel(b, a) <= 2e-6, max_rel(b, a)
    # the forwards differ from step to step (the weights moved), so a stale image would have shown
    assert max_rel(outs_g[2], outs_g[0]) > 1e-5


def test_eager_step_after_capture_keeps_the_graph_valid():
    """An un-captured step between replays (an odd-shaped last batch) rewrites the optimiser's pinned pointer table
    with its own gradient addresses; GraphedTrainStep.eager_step restores the captured table."""
    from paradis_model_amd.harness import GraphedTrainStep
    model_e, step_e, batches = _setup(False)
    model_g, step_g, _ = _setup(True)
    warm = 2
    g = GraphedTrainStep(step_g, batches[0], warmup=warm)
    for _ in range(warm):
        step_e(batches[0])
    small = tuple(t[:1].contiguous() for t in batches[1])
    for batch, graphed in ((batches[0], True), (small, False), (batches[1], True), (batches[0], True)):
        step_e(batch)
        if graphed:
            g(batch)
        else:
            g.eager_step(batch)
    torch.cuda.synchronize()
    pe = torch.cat([p.detach().flatten() for p in model_e.parameters()])
    pg = torch.cat([p.detach().flatten() for p in model_g.parameters()])
    assert max_rel(pg, pe) <= 1e-6, max_rel(pg, pe)


def test_eager_step_right_behind_a_gpu_bound_replay():
    """``graph.replay()`` is asynchronous and the captured optimiser node reads the pinned pointer table when the
    GPU reaches it.  With the queue ~200 ms deep (large fp32 matmuls in front of the replay stand in for a GPU-bound
    configuration: the 16x32 reduced model alone is host-bound) an ``eager_step`` issued right behind the replay used to
    overwrite the table under it - the replay then applied AdamW with the eager step's gradient addresses.
    ``eager_step`` now waits for the last replay (an event recorded after every replay).  Twin: the same steps eagerly."""
    from paradis_model_amd.harness import GraphedTrainStep
    model_e, step_e, batches = _setup(False)
    model_g, step_g, _ = _setup(True)
    warm = 2
    g = GraphedTrainStep(step_g, batches[0], warmup=warm)
    for _ in range(warm):
        step_e(batches[0])
    small = tuple(t[:1].contiguous() for t in batches[1])
    a = torch.randn(8192, 8192, device="cuda")
    for rnd in range(3):
        step_e(batches[rnd % 2]); step_e(small)
        torch.cuda.synchronize()
        for _ in range(20):            # ~10 ms each: the replay below sits behind them in the queue
            a @ a
        g(batches[rnd % 2])
        assert not g._replayed.query()             # the replay has not run yet: the race window is open
        g.eager_step(small)
    g(batches[0]); step_e(batches[0])
    torch.cuda.synchronize()
    pe = torch.cat([p.detach().flatten() for p in model_e.parameters()])
    pg = torch.cat([p.detach().flatten() for p in model_g.parameters()])
    assert max_rel(pg, pe) <= 1e-6, max_rel(pg, pe)
