"""Muon / NorMuon (row f3, second half).  PARITY UNPINNED: the reference takes these optimisers from
the un-vendored, un-pinned `dion` package; the oracle restates the published algorithm.  The CPU
tests check the oracle's defining properties and the host-side grouping; the GPU tests check the
HIP step against the oracle (fp32 on both sides, 2e-4 of the update's magnitude: five Newton-Schulz
iterations amplify GEMM summation-order differences)."""
import pytest
import torch

from oracle import muon_oracle as MO


def test_newton_schulz_orthogonalises():
    torch.manual_seed(0)
    for shape in ((48, 96), (96, 48), (64, 64)):
        G = torch.randn(*shape)
        X = MO.newton_schulz(G, 1e-8)
        s = torch.linalg.svdvals(X)
        assert X.shape == G.shape and float(s.max()) < 1.3 and float(s.min()) > 0.3, (shape, s.min(), s.max())
        # same singular vectors as G: U^T X V is (nearly) diagonal and positive
        U, _, Vh = torch.linalg.svd(G, full_matrices=False)
        D = U.T @ X @ Vh.T
        off = D - torch.diag(torch.diag(D))
        assert float(off.abs().max()) < 5e-3 and float(torch.diag(D).min()) > 0.3


def test_learning_rate_adjustment_and_decay():
    assert MO.adjusted_lr(0.1, (64, 16, 1, 1), "spectral_norm") == pytest.approx(0.2)
    assert MO.adjusted_lr(0.1, (16, 4, 5, 5), "rms_norm") == pytest.approx(0.1 * 0.2 * 10.0)
    assert MO.adjusted_lr(0.1, (8, 8), None) == 0.1
    W, G, M = torch.ones(4, 4), torch.zeros(4, 4), torch.zeros(4, 4)
    out = MO.muon_step(W, G, M, lr=0.5, weight_decay=0.1)      # zero gradient: only the decay acts
    assert torch.allclose(out, torch.full((4, 4), 0.95))


def test_param_groups_mirror_reference_rule():
    from torch import nn
    from paradis_model_amd.optim import build_param_groups
    m = nn.Sequential(nn.Conv2d(3, 4, 1), nn.Linear(4, 2, bias=False))
    m.register_parameter("alpha", nn.Parameter(torch.zeros(3)))
    groups = build_param_groups(m, 1e-3, 1e-2, "normuon")
    assert groups[0]["algorithm"] == "normuon" and groups[0]["flatten"] is True
    assert [tuple(p.shape) for p in groups[0]["params"]] == [(4, 3, 1, 1), (2, 4)]
    assert groups[1]["algorithm"] == "adamw"
    assert sorted(tuple(p.shape) for p in groups[1]["params"]) == [(3,), (4,)]


def test_abi_validation_without_gpu():
    from paradis_model_amd import _lib
    L = _lib.lib
    assert L.paradis_muon_step(None, 2, 2, 0, 4, 0.1, 0.1, 0.95, 0.95, 0.0, 1e-8, 0, 0, 0, None, None) == 1
    assert L.paradis_muon_step(None, 1, 2, 4, 4, 0.1, 0.1, 0.95, 0.95, 0.0, 1e-8, 0, 0, 0, None, None) == 1   # stride < T
    assert L.paradis_muon_step(None, 2, 2, 4, 4, 0.1, 0.1, 0.95, 0.95, 0.0, 1e-8, 0, 0, 0, None, None) == 1
    assert "workspace" in _lib.last_error()
    assert L.paradis_muon_step(None, 2, 0, 4, 4, 0.1, 0.1, 0.95, 0.95, 0.0, 1e-8, 0, 0, 0, None, None) == 0    # empty group
    assert L.paradis_muon_ws_bytes(3, 64, 48) >= 3 * (4 * 64 * 48 + 3 * 48 * 48) * 4
    assert L.paradis_bgemm(None, None, None, None, 1, 0, 4, 4, 0, 0, 0, 0, None, None) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("cls_name,nesterov", [("Muon", False), ("Muon", True), ("NorMuon", False)])
@pytest.mark.parametrize("split", [True, False])
def test_hip_step_vs_oracle(cls_name, nesterov, split, monkeypatch):
    """Both arithmetics of the Newton-Schulz products (bf16-split / exact f32 MFMA) against the oracle."""
    from paradis_model_amd import ops, optim
    monkeypatch.setattr(ops, "GEMM_SCHEME", ops.GEMM_BF16X3 if split else ops.GEMM_EXACT)
    torch.manual_seed(1)
    # two matrices share a shape (stacked in one launch); wide, tall, conv, depthwise, ragged
    shapes = [(64, 48), (48, 64), (64, 48), (32, 16, 1, 1), (40, 1, 3, 3), (20, 8), (130, 258)]
    params = [torch.randn(*s) * 0.1 for s in shapes]
    extra = torch.randn(17)                                       # an AdamW-group parameter
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in params]
    mine_extra = torch.nn.Parameter(extra.clone().cuda())
    cls = getattr(optim, cls_name)
    adjust = "rms_norm" if cls_name == "NorMuon" else "spectral_norm"
    opt = cls([dict(params=mine, algorithm=cls_name.lower(), flatten=True),
               dict(params=[mine_extra], algorithm="adamw")],
              lr=5e-3, weight_decay=1e-2, betas=(0.9, 0.95), nesterov=nesterov, use_triton=True)
    ref_extra = torch.nn.Parameter(extra.clone())
    ref_adam = torch.optim.AdamW([ref_extra], lr=5e-3, weight_decay=1e-2, betas=(0.9, 0.95))
    W = [p.clone() for p in params]
    M = [torch.zeros_like(p) for p in params]
    V = [torch.zeros(p.shape[0], 1) for p in params] if cls_name == "NorMuon" else [None] * len(params)
    for step in range(3):
        grads = [torch.randn_like(p) * (0.5 + step) for p in params]
        ge = torch.randn_like(extra)
        for m, g in zip(mine, grads):
            m.grad = g.clone().cuda()
        mine_extra.grad = ge.clone().cuda()
        ref_extra.grad = ge.clone()
        opt.step()
        ref_adam.step()
        for i, g in enumerate(grads):
            Wn = MO.muon_step(W[i], g, M[i], lr=5e-3, weight_decay=1e-2, nesterov=nesterov, adjust=adjust, V=V[i])
            upd = (Wn - W[i]).abs().max()
            err = (mine[i].detach().cpu() - Wn).abs().max()
            assert float(err) <= 2e-4 * float(upd) + 1e-7, (cls_name, shapes[i], step, float(err), float(upd))
            W[i] = Wn
        assert torch.allclose(mine_extra.detach().cpu(), ref_extra.detach(), rtol=1e-6, atol=1e-7)
    st = opt.state[mine[0]]
    assert "momentum" in st and (("variance_neuron" in st) == (cls_name == "NorMuon"))
