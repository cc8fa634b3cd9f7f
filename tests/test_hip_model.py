"""GPU parity of the full drop-in model against reference goldens (G4, c2) and the CPU oracle."""
import pytest
import torch

from oracle import paradis_oracle as O
from paradis_model_amd.config import default_config, feature_layout, reduced_config, stub_datamodule
from tests._util import assert_chk, load_golden, make_grid, max_rel, seeded

pytestmark = pytest.mark.gpu


def _build(cfg, lg, og, state=None):
    from paradis_model_amd.model import Paradis
    torch.manual_seed(42)
    m = Paradis(stub_datamodule(cfg), cfg, lg, og)
    if state is not None:
        m.load_state_dict(state, strict=True)
    return m.cuda()


@pytest.mark.parametrize("variant,fuse_projection", [("a", False), ("b", False), ("c", False), ("a", True),
                                                     ("c", True)])
@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3", "exact"])
def test_reduced_model_vs_reference_golden(variant, fuse_projection, gemm, monkeypatch):
    """Same tolerances for the three GEMM arithmetics (bf16x3 = default, exact f32 MFMA, opt-in f16x2)."""
    from paradis_model_amd import ops
    from paradis_model_amd.loss import build_loss
    from paradis_model_amd.model import blocks
    monkeypatch.setattr(ops, "GEMM_SCHEME", ops._SCHEMES[gemm])
    # the large-plane path (GlobalBias projection inside the GEMM) must give the same model
    monkeypatch.setattr(blocks, "FUSE_BIAS_PROJECTION_MIN_POINTS", 0 if fuse_projection else 1 << 40)
    rec = load_golden(f"g4_model_{variant}.pt")
    v = rec["variant"]
    cfg = reduced_config(activation=v["activation"], adv_interpolation=v["adv_interpolation"],
                         coarsening_factor=v["coarsening_factor"])
    lg, og = rec["lat_grid"], rec["lon_grid"]
    model = _build(cfg, lg, og, rec["state"])
    x = seeded(rec["x_seed"], rec["B"], 186, v["nlat"], v["nlon"])
    x[:, -2] = lg
    x[:, -1] = og
    tgt = seeded(rec["target_seed"], rec["B"], 97, v["nlat"], v["nlon"])
    assert_chk([x, tgt], rec["chk"])
    xd = x.cuda().requires_grad_(True)
    y = model(xd)
    e = max_rel(y.detach().cpu(), rec["y"])
    assert e <= 1e-5, e                                   # north_star: 1e-5 relative
    loss_fn = build_loss(cfg, rec["lat_deg"]).cuda()
    loss = loss_fn(y, tgt.cuda())
    assert abs(float(loss) - float(rec["loss"])) <= 2e-6 * abs(float(rec["loss"]))
    loss.backward()
    e_gx = max_rel(xd.grad.cpu()[:, ::9], rec["gx_sub"])
    print("MEASURED reduced model %s %s: forward %.2e, input gradient %.2e" % (variant, gemm, e, e_gx))
    assert e_gx <= GX_BOUND[variant], e_gx
    worst = ("", 0.0)
    for n, p in model.named_parameters():
        gref = rec["grads"][n]
        if float(gref.abs().max()) == 0:
            continue
        err = max_rel(p.grad.cpu(), gref)
        if err > worst[1]:
            worst = (n, err)
    print("worst grad", worst)
    # measured (round 4): variant a 4.4-5.0e-4 (one bias of the velocity network, whose gradient is a sum over the
    # ill-conditioned points next to the poles; same figure in all three GEMM arithmetics), b 1.1-1.2e-4, c 2.5-5.3e-5
    assert worst[1] <= {"a": 8e-4, "b": 2e-4, "c": 1e-4}[variant], worst


# Bounds = at most 2 x what the shipped library measures against the REFERENCE's own CPU-fp32 goldens (round 4, the
# MEASURED lines these tests print; INTEGRATION.md "Tolerances" quotes the same figures):
#   input gradient of the reduced models 2.7e-7 .. 7.4e-7 in all three arithmetics (rounds 1-3 asserted 5e-4);
#   two-step rollout: outputs 7.8e-7 / 1.0e-6, alpha_adv gradient 5.9e-7, largest gradient-norm deviation 8.6e-6
#   (rounds 1-3: 2e-5 / 1e-3 / 2e-3).
GX_BOUND = {"a": 1.5e-6, "b": 1.5e-6, "c": 1.5e-6}
ROLLOUT_BOUNDS = {"outputs": 2.0e-6, "alpha": 1.2e-6, "norms": 1.7e-5}


def test_two_step_rollout_vs_reference_golden():
    from paradis_model_amd.loss import build_loss
    rec = load_golden("c2_rollout.pt")
    cfg = reduced_config()
    model = _build(cfg, rec["lat_grid"], rec["lon_grid"], rec["state"])
    H, W = rec["lat_grid"].shape
    B, S = rec["B"], rec["S"]
    s = rec["seeds"]
    inp = seeded(s[0], B, 1, 166, H, W).cuda()
    tgt = seeded(s[1], B, S, 97, H, W).cuda()
    forc = seeded(s[2], B, S, H, W, 10, kind="rand").cuda()
    const = seeded(s[3], B, 1, H, W, 10).cuda()
    loss_fn = build_loss(cfg, rec["lat_deg"]).cuda()
    from paradis_model_amd.harness import rollout_loss
    total, outs = rollout_loss(model, loss_fn, (inp, tgt, forc, const), num_common=83, n_inputs=2,
                               keep_outputs=True)
    e_out = [max_rel(got.detach().cpu(), want) for got, want in zip(outs, rec["outputs"])]
    e_alpha = max_rel(model.alpha_adv.grad.cpu(), rec["grad_alpha"])
    e_norm = max((abs(float(p.grad.norm()) - rec["grad_norms"][n]) / max(rec["grad_norms"][n], 1e-30), n)
                 for n, p in model.named_parameters())
    print("MEASURED two-step rollout: outputs %s, alpha_adv gradient %.2e, worst gradient-norm deviation %.2e (%s)"
          % (["%.2e" % x for x in e_out], e_alpha, e_norm[0], e_norm[1]))
    # bounds = 2 x what the shipped library measures ("Tolerances"): see ROLLOUT_BOUNDS
    assert max(e_out) <= ROLLOUT_BOUNDS["outputs"], e_out
    assert abs(float(total) - float(rec["loss"])) <= 2e-6 * abs(float(rec["loss"]))
    assert e_alpha <= ROLLOUT_BOUNDS["alpha"], e_alpha
    assert e_norm[0] <= ROLLOUT_BOUNDS["norms"], e_norm


def test_default_config_forward_vs_oracle():
    """Full-size (60 M parameter) model at reference init, 32x64, B=1: HIP vs CPU oracle."""
    cfg = default_config()
    _, lg, og = make_grid(32, 64, False)
    model = _build(cfg, lg, og)
    with torch.no_grad():  # make the bias maps / gates non-trivial
        g = torch.Generator().manual_seed(7)
        for n, p in model.named_parameters():
            if n.endswith((".A", ".U", ".V")):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    lay = feature_layout(cfg)
    spec = O.spec_from_cfg(cfg, 32, 64, lay.num_in_dyn_features, lay.num_in_static_features,
                           lay.num_out_features)
    x = seeded(5, 1, 186, 32, 64)
    x[:, -2], x[:, -1] = lg, og
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.paradis_forward(params, spec, x, lg, og, interp_impl="aten_ref")
        got = model(x.cuda()).cpu()
    e = max_rel(got, want)
    print("default-config forward max-rel", e)
    assert e <= 1e-5, e


def _rms_rel(a, b):
    d = a.double() - b.double()
    return float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-30))


def _oracle_grads(model, spec, x, ct, lg, og, dtype):
    ps = {k: v.detach().cpu().to(dtype).requires_grad_(True) if v.dtype.is_floating_point else v.detach().cpu()
          for k, v in model.state_dict().items()}
    # (ATen's grid_sample in both precisions: in fp64 it agrees with the oracle's explicit tap formula to 1e-12 -
    #  tests/test_oracle_golden.py - and runs 30 x faster on the host)
    y = O.paradis_forward(ps, spec, x.to(dtype), lg.to(dtype), og.to(dtype), interp_impl="aten_ref")
    (y * ct.to(dtype)).sum().backward()
    return y.detach(), {k: v.grad for k, v in ps.items() if torch.is_tensor(v) and v.requires_grad}


def _check_grads_by_fp64_protocol(model, g32, g64, factor=5.0, floor=2e-5, rms_factor=4.0, rms_floor=5e-5):
    """the HIP gradient's distance to the fp64 oracle against the CPU fp32 oracle's own
    distance (the velocity path amplifies fp32 coordinate rounding; a few ill-conditioned points
    near the poles decide the maximum of a weight gradient).  Measured on the default model
    (tools/grad_probe.py, round 2): over all 335 parameters the HIP error is 2.4x the CPU-fp32 error in the
    median; the largest ratios (up to 10) belong to gradients whose absolute error is below 2e-5 (the
    floor), the largest ratio among the others is 5 (velocity networks: 3.6e-4 where the CPU has 7e-5).  The factor is the accumulation order: an MFMA accumulator takes the K
    products of a dot product one after the other (512 updates for K = 1024), the CPU's vector units
    keep 16 partial sums per accumulator - sqrt(512/64) = 2.8.
    Round 4: with the sign checkerboard of the bf16x3 GEMMs (the offset every output shared is gone from the sums over
    pixels and channels) the largest ratios above the floors are 2.8 (max-abs) and 2.1 (norm-wise) on the default
    model at 32x64, 1.8 / 1.2 on the reduced model at 128x256 and 65x130 - the factors are now 5 and 4 (rounds 2-3: 8
    and 5, with 5.0-6.8 measured)."""
    worst = ("", 0.0, 0.0)
    bad = []
    top_m, top_r = (0.0, ""), (0.0, "")
    for n, p in model.named_parameters():
        ref = g64.get(n)
        if ref is None or float(ref.abs().max()) == 0:
            continue
        e_gpu = max_rel(p.grad.cpu().double(), ref)
        e_cpu = max_rel(g32[n].double(), ref)
        # Two statistics.  max-abs / max-abs: a ratio of two maxima of heavy-tailed point errors - it
        # moves by 2x from one rounding pattern to the next (round 2, f16x2 GEMMs: largest ratio 5; round 3, bf16x3
        # GEMMs and the restated advection map: 6.8 on velocity_nets.4.1-SepConv.pointwise.weight at an error of
        # 3e-3, 4.4-4.8 on the velocity networks' norm scales at 5e-4) - hence factor 8.  Norm-wise error: the
        # stable statistic, factor 5 over a floor of 5e-5 (measured: median ratio 2.4; 5.7 only on
        # static_encoder.0.depthwise.weight whose error is 3e-5).
        r_gpu, r_cpu = _rms_rel(p.grad.cpu().double(), ref), _rms_rel(g32[n].double(), ref)
        if not (e_gpu <= factor * e_cpu + floor and r_gpu <= rms_factor * r_cpu + rms_floor):
            bad.append((n, e_gpu, e_cpu, r_gpu, r_cpu))
        if e_gpu > worst[1]:
            worst = (n, e_gpu, e_cpu)
        # what the bounds are measured against: the largest ratios among the gradients above the floors
        if e_gpu > floor:
            top_m = max(top_m, ((e_gpu - floor) / max(e_cpu, 1e-30), n))
        if r_gpu > rms_floor:
            top_r = max(top_r, ((r_gpu - rms_floor) / max(r_cpu, 1e-30), n))
    print("MEASURED fp64 protocol: largest (max-abs error - floor) / cpu32 error %.2f (%s); largest (norm-wise error - "
          "floor) / cpu32 error %.2f (%s); bounds %.0f / %.0f" % (top_m[0], top_m[1], top_r[0], top_r[1], factor, rms_factor))
    assert not bad, bad
    return worst


def test_default_config_backward_vs_oracle():
    """Full-size model, 32x64, B=2: every parameter gradient of the HIP path (LDS-DMA GEMMs with
    split-k weight gradients, stream-twice ChannelNorm backward with the fused residual gradient,
    whole-plane advection scatter, 16-byte stencil staging) by the fp64 protocol."""
    cfg = default_config()
    _, lg, og = make_grid(32, 64, False)
    model = _build(cfg, lg, og)
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        for n, p in model.named_parameters():
            if n.endswith((".A", ".U", ".V")):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    lay = feature_layout(cfg)
    spec = O.spec_from_cfg(cfg, 32, 64, lay.num_in_dyn_features, lay.num_in_static_features,
                           lay.num_out_features)
    x = seeded(5, 2, 186, 32, 64)
    x[:, -2], x[:, -1] = lg, og
    ct = seeded(6, 2, lay.num_out_features, 32, 64)
    y32, g32 = _oracle_grads(model, spec, x, ct, lg, og, torch.float32)
    _, g64 = _oracle_grads(model, spec, x, ct, lg, og, torch.float64)
    got = model(x.cuda())
    (got * ct.cuda()).sum().backward()
    assert max_rel(got.detach().cpu(), y32) <= 1e-5
    worst = _check_grads_by_fp64_protocol(model, g32, g64)
    print("default-config worst grad error vs fp64 (gpu, cpu32)", worst)


@pytest.mark.parametrize("nlat,nlon,poles", [(128, 256, False), (65, 130, True), (9, 30, True), (20, 50, False)])
def test_reduced_model_on_large_grids_vs_oracle(nlat, nlon, poles):
    """(The last two: ragged small grids - W not a multiple of 4, so every 16-byte staging path falls back; an odd
    latitude count with pole rows.)  The large-plane code paths inside the model - tiled advection windows (forward and backward),
    GlobalBias projection inside the GEMM epilogue (planes >= 8192 points), multi-tile stencils -
    against the CPU oracle, forward and every parameter gradient.  Gradients: the velocity path
    amplifies fp32 coordinate rounding with the grid size (SURVEY 8c iii), hence 2e-3 there."""
    cfg = reduced_config()
    _, lg, og = make_grid(nlat, nlon, poles)
    model = _build(cfg, lg, og)
    with torch.no_grad():
        g = torch.Generator().manual_seed(11)
        for n, p in model.named_parameters():
            if n.endswith((".A", ".U", ".V")):
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    lay = feature_layout(cfg)
    spec = O.spec_from_cfg(cfg, nlat, nlon, lay.num_in_dyn_features, lay.num_in_static_features,
                           lay.num_out_features)
    x = seeded(9, 1, 186, nlat, nlon)
    x[:, -2], x[:, -1] = lg, og
    ct = seeded(10, 1, lay.num_out_features, nlat, nlon)
    y32, g32 = _oracle_grads(model, spec, x, ct, lg, og, torch.float32)
    _, g64 = _oracle_grads(model, spec, x, ct, lg, og, torch.float64)
    got = model(x.cuda())
    (got * ct.cuda()).sum().backward()
    e_fwd = max_rel(got.detach().cpu(), y32)
    print("large-grid reduced model forward max-rel", nlat, nlon, e_fwd)
    assert e_fwd <= 1e-5, e_fwd
    worst = _check_grads_by_fp64_protocol(model, g32, g64)
    print("large-grid worst grad error vs fp64 (gpu, cpu32)", nlat, nlon, worst)


def test_gradient_checkpointing_matches():
    cfg = reduced_config()
    _, lg, og = make_grid(16, 32, False)
    x = seeded(3, 2, 186, 16, 32).cuda()
    grads = []
    for ck in (False, True):
        cfg.compute.gradient_checkpointing = ck
        m = _build(cfg, lg, og)
        m(x).square().mean().backward()
        grads.append(torch.cat([p.grad.flatten() for p in m.parameters()]).cpu())
    assert max_rel(grads[1], grads[0]) <= 1e-5


@pytest.mark.parametrize("gemm", ["bf16x3", "f16x2", "exact"])
def test_forward_under_inference_mode(gemm, monkeypatch):
    """Lightning runs validation / sanity-check / predict steps under ``torch.inference_mode()`` (reference
    train.py:44, forecast.py:99 leave ``inference_mode`` at its default): parameters moved or created there and
    every op output are inference tensors, which have no version counter.  Same values as under ``no_grad``."""
    from paradis_model_amd import ops
    monkeypatch.setattr(ops, "GEMM_SCHEME", ops._SCHEMES[gemm])
    cfg = reduced_config()
    _, lg, og = make_grid(16, 32, False)
    model = _build(cfg, lg, og)
    x = seeded(3, 2, 186, 16, 32).cuda()
    with torch.no_grad():
        want = model(x)
    with torch.inference_mode():
        got = model(x)
        got2 = model(x.clone())          # an inference-tensor input as well
        w = model.input_proj[0].conv.weight.clone()      # an inference-tensor weight: no version counter
        y = ops.pointwise(x, w)
    assert torch.equal(got, want) and torch.equal(got2, want)
    assert y.shape == (2, w.shape[0], 16, 32) and bool(torch.isfinite(y).all())


@pytest.mark.parametrize("H,W", [(16, 32), (32, 64)])
def test_empty_batch_forward_and_backward(H, W):
    """B = 0 (the last, empty shard of an uneven distributed sampler; a filtered-out batch): every op returns an empty
    tensor of the right shape without launching on zero-sized grids, the backward gives every parameter a zero
    gradient, and a training step on it leaves the library usable (the next real step matches a fresh run)."""
    cfg = reduced_config()
    _, lg, og = make_grid(H, W, False)
    model = _build(cfg, lg, og)
    x0 = torch.zeros(0, 186, H, W, device="cuda", requires_grad=True)
    y0 = model(x0)
    assert tuple(y0.shape) == (0, 97, H, W)
    y0.sum().backward()
    assert tuple(x0.grad.shape) == (0, 186, H, W)
    for n, p in model.named_parameters():
        assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
    # the library still works afterwards, and the empty step has changed nothing
    x = seeded(3, 2, 186, H, W).cuda()
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        got = model(x)
    twin = _build(cfg, lg, og)
    with torch.no_grad():
        want = twin(x)
    assert torch.equal(got, want)
