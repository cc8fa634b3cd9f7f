"""GPU parity of the 'next' rows f1-f3: fused ParadisLoss, channel-block copies, AdamW kernel."""
import pytest
import torch

from oracle import paradis_oracle as O
from paradis_model_amd.config import default_config
from tests._util import load_golden, max_rel, seeded

pytestmark = pytest.mark.gpu


def test_fused_loss_vs_reference_golden():
    from paradis_model_amd.loss import build_loss
    g = load_golden("g6_loss.pt")
    cfg = default_config()
    for key, rec in g.items():
        if "loss" not in rec:
            continue
        nlat = int(key.split("x")[0]); nlon = int(key.split("x")[1].split("_")[0])
        cfg.training.loss_function.type = key.split("_", 1)[1]
        fn = build_loss(cfg, rec["lat_deg"]).cuda()
        p = seeded(rec["pred_seed"], 2, 97, nlat, nlon, scale=1.5).cuda().requires_grad_(True)
        t = seeded(rec["target_seed"], 2, 97, nlat, nlon).cuda()
        loss = fn(p, t)
        (loss * 0.5).backward()                      # exercises the upstream-gradient scaling
        assert abs(float(loss) - float(rec["loss"])) <= 2e-6 * abs(float(rec["loss"])), key
        assert max_rel(2.0 * p.grad.cpu()[:, ::8, ::2, ::4], rec["gpred_sub"]) <= 1e-5, key


def test_concat_channels_and_grad():
    from paradis_model_amd import ops
    a, b, c = (seeded(i, 2, n, 8, 16) for i, n in ((1, 5), (2, 3), (3, 4)))
    big = seeded(9, 2, 12, 8, 16)
    parts = [a.cuda().requires_grad_(True), big.cuda()[:, 2:5], c.cuda().requires_grad_(True)]
    out = ops.concat_channels(parts)
    want = torch.cat([a, big[:, 2:5], c], 1)
    assert torch.equal(out.cpu(), want)
    ct = seeded(4, *want.shape)
    out.backward(ct.cuda())
    assert torch.equal(parts[0].grad.cpu(), ct[:, :5]) and torch.equal(parts[2].grad.cpu(), ct[:, 8:])


def test_adamw_kernel_matches_torch():
    from paradis_model_amd.optim import AdamW
    torch.manual_seed(0)
    # several tensors (one launch per group), one longer than a chunk of the fused kernel
    ps = [torch.randn(1000), torch.randn(37, 5), torch.randn(3), torch.randn(70001)]
    gs = [[torch.randn_like(p) * (10.0 ** (i - 1)) for p in ps] for i in range(4)]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    kw = dict(lr=5e-4, weight_decay=1e-2, betas=(0.9, 0.95))
    o_ref, o_mine = torch.optim.AdamW(ref, **kw), AdamW(mine, **kw)
    for step in range(4):
        for i, (r, m, gq) in enumerate(zip(ref, mine, gs[step])):
            if step == 2 and i == 1:      # a parameter without a gradient: its step count falls behind
                r.grad = m.grad = None    # and the group takes the per-tensor path from then on
                continue
            r.grad, m.grad = gq.clone(), gq.clone().cuda()
        o_ref.step(); o_mine.step()
    for r, m in zip(ref, mine):
        assert max_rel(m.detach().cpu(), r.detach()) <= 1e-6
    sd = o_mine.state_dict()["state"][0]
    assert set(sd) == {"step", "exp_avg", "exp_avg_sq"}


def test_adamw_state_dict_round_trip_between_steps():
    """load_state_dict after a fused step replaces the moment tensors behind the same parameter ids:
    the next step must update the LOADED moments (the address table is rebuilt every step)."""
    import copy
    from paradis_model_amd.optim import AdamW
    torch.manual_seed(1)
    ps = [torch.randn(300), torch.randn(17, 9), torch.randn(5000)]
    gs = [[torch.randn_like(p) for p in ps] for _ in range(3)]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    kw = dict(lr=1e-2, weight_decay=1e-2, betas=(0.9, 0.95))
    o_ref, o_mine = torch.optim.AdamW(ref, **kw), AdamW(mine, **kw)

    def step(k):
        for r, m, g in zip(ref, mine, gs[k]):
            r.grad, m.grad = g.clone(), g.clone().cuda()
        o_ref.step(); o_mine.step()

    step(0)
    sd_ref, sd_mine = copy.deepcopy(o_ref.state_dict()), copy.deepcopy(o_mine.state_dict())
    step(1)                                   # moves the moments away from the snapshot
    o_ref.load_state_dict(sd_ref)             # back to the snapshot: new state tensors, same parameter ids
    o_mine.load_state_dict(sd_mine)
    old_ptrs = {id(p): o_mine.state[p]["exp_avg"].data_ptr() for p in mine}
    step(2)
    for r, m in zip(ref, mine):
        assert max_rel(m.detach().cpu(), r.detach()) <= 1e-6
        assert max_rel(o_mine.state[m]["exp_avg"].cpu(), o_ref.state[r]["exp_avg"]) <= 1e-6
        assert o_mine.state[m]["exp_avg"].data_ptr() == old_ptrs[id(m)]
    # a parameter re-homed with .data = (model.to(), manual re-allocation) is followed as well
    with torch.no_grad():
        mine[0].data = mine[0].data.clone()
    step(0)
    for r, m in zip(ref, mine):
        assert max_rel(m.detach().cpu(), r.detach()) <= 1e-6
