"""Every BASELINE.json config on the GPU (`-m gpu`): configs[2] (32x64, 6-step rollout, B=32),
configs[3]'s shape (128x256, default model) and configs[4] (721x1440 with pole rows, forward).

Sizes the CPU oracle finishes in seconds are compared with the oracle (1e-5 forward, north_star);
the full sizes are covered by size-independent properties of the operator:
  * longitude-roll equivariance - rolling every longitude-indexed quantity (input columns,
    GlobalBias V) by s columns rolls the output by s columns: the departure longitude is
    lon_a(x) + f(u, v, lat_a), so dep'(x) = dep(x - s) + s*dlon (mod 2 pi) and the sample lands on the
    rolled field's rolled cell (reference model/advection.py:129-169, model/padding.py:26-37);
  * the autoregressive loss of a B=32 batch equals the mean of its per-sample losses
    (reference trainer.py:508-576: samples never interact).
"""
import pytest
import torch

from oracle import paradis_oracle as O
from paradis_model_amd.config import default_config, feature_layout, reduced_config, stub_datamodule
from tests._util import make_grid, max_rel, rms_rel, seeded

pytestmark = pytest.mark.gpu


def _build(cfg, lg, og, seed=42, bias_scale=0.2):
    from paradis_model_amd.model import Paradis
    torch.manual_seed(seed)
    m = Paradis(stub_datamodule(cfg), cfg, lg, og)
    with torch.no_grad():  # non-trivial GlobalBias maps (reference init is N(0, 1e-3))
        g = torch.Generator().manual_seed(7)
        for n, p in m.named_parameters():
            if n.endswith((".A", ".U", ".V")):
                p.copy_(torch.randn(p.shape, generator=g) * bias_scale)
    return m.cuda()


def _spec(cfg, H, W):
    lay = feature_layout(cfg)
    return O.spec_from_cfg(cfg, H, W, lay.num_in_dyn_features, lay.num_in_static_features,
                           lay.num_out_features)


def _loss_weights(cfg, lat_deg):
    fw = O.feature_weights(torch.tensor([1.0] * 83 + [0.1] * 13 + [1.0]),
                           torch.tensor(cfg.features.pressure_levels), 97, 6)
    return fw, O.latitude_weights(lat_deg)


def _batch(B, S, H, W, lg, og, seed0=100):
    inp = seeded(seed0, B, 1, 166, H, W)
    tgt = seeded(seed0 + 1, B, S, 97, H, W)
    forc = seeded(seed0 + 2, B, S, H, W, 10, kind="rand")
    const = seeded(seed0 + 3, B, 1, H, W, 10)
    const[..., -2], const[..., -1] = lg, og
    return inp, tgt, forc, const


# ------------------------------------------------------------------------------------------------
# configs[2]: 32x64, 6 autoregressive steps
# ------------------------------------------------------------------------------------------------
def test_cfg2_six_step_rollout_default_model_vs_oracle():
    """Default 60 M-parameter model, S=6, B=1: per-step outputs, loss and the gate gradient against the
    oracle's restated training loop (reference trainer.py:508-576, 710-729)."""
    from paradis_model_amd.harness import rollout_loss
    from paradis_model_amd.loss import build_loss
    cfg = default_config()
    H, W, S, B = 32, 64, 6, 1
    lat_deg, lg, og = make_grid(H, W, False)
    model = _build(cfg, lg, og)
    spec = _spec(cfg, H, W)
    inp, tgt, forc, const = _batch(B, S, H, W, lg, og)
    # oracle loop
    P = {k: (v.detach().cpu().clone().requires_grad_(True) if v.dtype.is_floating_point else v.detach().cpu())
         for k, v in model.state_dict().items()}
    fw, lw = _loss_weights(cfg, lat_deg)
    constants = const[:, :1].permute(0, 1, 4, 2, 3)
    forcings = forc.permute(0, 1, 4, 2, 3)
    cur, total, want_outs = inp, 0.0, []
    for step in range(S):
        mi = torch.cat([cur, forcings[:, step].unsqueeze(1), constants], dim=2).squeeze(1)
        y = O.paradis_forward(P, spec, mi, lg, og, interp_impl="aten_ref")
        want_outs.append(y.detach())
        total = total + O.paradis_loss(y, tgt[:, step], fw, lw) / S
        cur = torch.cat([mi[:, 83:166], y[:, :83]], dim=1).unsqueeze(1)
    total.backward()
    # HIP path
    loss_fn = build_loss(cfg, lat_deg).cuda()
    got_total, outs = rollout_loss(model, loss_fn, tuple(t.cuda() for t in (inp, tgt, forc, const)),
                                   num_common=83, n_inputs=2, keep_outputs=True)
    errs = [max_rel(g.cpu(), w) for g, w in zip(outs, want_outs)]
    print("cfg2 per-step forward max-rel", ["%.1e" % e for e in errs])
    # every one of the six steps at the 1e-5 bar (measured 2.0-2.4e-6 at each step)
    assert max(errs) <= 5e-6, errs           # measured 1.9-2.3e-6 at each step (north star: 1e-5)
    assert abs(float(got_total) - float(total)) <= 5e-6 * abs(float(total))
    ga = model.alpha_adv.grad.cpu()
    e_alpha = max_rel(ga, P["alpha_adv"].grad)
    print("cfg2 alpha_adv grad max-rel", e_alpha)
    assert e_alpha <= 2e-6, e_alpha          # measured 7.3e-7 (round 3 asserted 1e-4)
    worst = ("", 0.0)
    for n, p in model.named_parameters():
        want = float(P[n].grad.norm())
        rel = abs(float(p.grad.norm()) - want) / (want + 1e-30)
        if rel > worst[1]:
            worst = (n, rel)
        assert abs(float(p.grad.norm()) - want) <= 1.4e-4 * want + 1e-9, (n, rel)      # measured 6.7e-5 (round 3 asserted 5e-4)
    print("cfg2 worst parameter-gradient norm deviation", worst)


def test_cfg2_full_batch_rollout_properties():
    """configs[2] at full size (B=32, S=6, default model, no checkpointing): the step runs inside the
    288 GB of one MI355X, loss and gradients are finite, and the batch loss equals the mean of the
    losses of its samples run one by one."""
    from paradis_model_amd.harness import rollout_loss
    from paradis_model_amd.loss import build_loss
    cfg = default_config()
    H, W, S, B = 32, 64, 6, 32
    lat_deg, lg, og = make_grid(H, W, False)
    model = _build(cfg, lg, og)
    loss_fn = build_loss(cfg, lat_deg).cuda()
    batch = tuple(t.cuda() for t in _batch(B, S, H, W, lg, og, seed0=200))
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    total, _ = rollout_loss(model, loss_fn, batch, num_common=83, n_inputs=2)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    print("cfg2 B=32 S=6 peak HBM %.1f GB, loss %.6f" % (peak / 1e9, float(total)))
    assert peak < 288e9
    assert torch.isfinite(total)
    gnorm = torch.stack([p.grad.norm() for p in model.parameters()])
    assert bool(torch.isfinite(gnorm).all()) and float(gnorm.max()) > 0
    # per-sample losses (forward only) for a few samples + linearity of the mean
    with torch.no_grad():
        per = []
        for b in range(B):
            one = tuple(t[b:b + 1] for t in batch)
            lb, _ = rollout_loss(model, loss_fn, one, num_common=83, n_inputs=2, backward=False)
            per.append(float(lb))
    mean = sum(per) / B
    assert abs(mean - float(total)) <= 2e-6 * abs(mean), (mean, float(total))
    del model, batch, loss_fn
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# configs[3] shape: 128x256
# ------------------------------------------------------------------------------------------------
def test_cfg3_default_model_128x256_forward_vs_oracle():
    """Default model on the 1.4 degree grid, B=1 (tiled advection windows, GlobalBias projection inside
    the GEMM epilogue, multi-tile stencils at full width) against the oracle."""
    cfg = default_config()
    H, W = 128, 256
    _, lg, og = make_grid(H, W, False)
    model = _build(cfg, lg, og)
    x = seeded(21, 1, 186, H, W)
    x[:, -2], x[:, -1] = lg, og
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.paradis_forward(params, _spec(cfg, H, W), x, lg, og, interp_impl="aten_ref")
        got = model(x.cuda()).cpu()
    e = max_rel(got, want)
    print("cfg3 128x256 default-model forward max-rel", e)
    assert e <= 1e-5, e


def _smooth(x, k=9, times=2):
    """low-pass in both directions (longitude periodic): what ERA5 fields look like next to white noise"""
    import torch.nn.functional as F
    h = k // 2
    for _ in range(times):
        x = F.pad(x, (h, h, 0, 0), mode="circular")
        x = F.pad(x, (0, 0, h, h), mode="replicate")
        x = F.avg_pool2d(x, k, stride=1)
    return x


def _oracle_grads_ckpt(model, spec, x, ct, lg, og, dtype):
    """oracle forward + every parameter gradient with one ADR layer's intermediates alive at a time.  Both
    precisions go through ATen's grid_sample (the reference's own operator; in fp64 it agrees with the oracle's
    explicit tap formula to 1e-12, tests/test_oracle_golden.py, and is 30 x faster on the host)."""
    ps = {k: v.detach().cpu().to(dtype).requires_grad_(True) if v.dtype.is_floating_point else v.detach().cpu()
          for k, v in model.state_dict().items()}
    y = O.paradis_forward(ps, spec, x.to(dtype), lg.to(dtype), og.to(dtype), interp_impl="aten_ref",
                          checkpoint_layers=True)
    (y * ct.to(dtype)).sum().backward()
    return y.detach(), {k: v.grad for k, v in ps.items() if torch.is_tensor(v) and v.requires_grad}


def test_cfg3_one_layer_128x256_gradients_single_seed():
    """The deterministic companion of the distributional test below: ONE ADR layer of the full-width model at
    128x256, one fixed seed, forward and every parameter gradient by the fp64 protocol at the single-seed bounds
    (8 x max-abs, 5 x norm-wise of the CPU-fp32 oracle's own distance to fp64) for BOTH arithmetics - the default bf16x3
    and the f32-MFMA kernels - against one pair of host oracle runs.  With one layer the chain amplifies rounding far less
    than with two (no advection differentiating through another layer's output), so a hard per-parameter assertion holds
    here where the two-layer case needs a distribution; a regression confined to one arithmetic or one seed of the
    distributional test still fails this one."""
    from paradis_model_amd import ops
    from tests.test_hip_model import _check_grads_by_fp64_protocol
    cfg = default_config()
    cfg.model.num_layers = 1
    H, W = 128, 256
    _, lg, og = make_grid(H, W, False)
    spec = _spec(cfg, H, W)
    model = _build(cfg, lg, og, bias_scale=0.05)
    x = _smooth(seeded(123, 1, 186, H, W)) * 4.0
    x[:, -2], x[:, -1] = lg, og
    ct = _smooth(seeded(124, 1, 97, H, W)) * 4.0
    y32, g32 = _oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float32)
    y64, g64 = _oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float64)
    keep = ops.GEMM_SCHEME
    try:
        for name in ("bf16x3", "exact"):
            ops.GEMM_SCHEME = ops._SCHEMES[name]
            model.zero_grad(set_to_none=True)
            got = model(x.cuda())
            (got * ct.cuda()).sum().backward()
            e = max_rel(got.detach().cpu(), y32)
            assert e <= 1e-5, (name, e)
            _check_grads_by_fp64_protocol(model, g32, g64, factor=8.0, rms_factor=5.0)
            r = rms_rel(model.get_parameter("reaction.0.0-ChannelNorm.bias").grad.cpu().double(),
                        g64["reaction.0.0-ChannelNorm.bias"]) / max(rms_rel(g32["reaction.0.0-ChannelNorm.bias"].double(),
                                                                            g64["reaction.0.0-ChannelNorm.bias"]), 1e-12)
            print("cfg3 128x256 L=1 %s: forward %.2e, ChannelNorm.bias ratio %.2f" % (name, e, r))
            assert r <= 5.0, (name, r)          # the pixel-sum gradient that exposed the bf16 MFMA offset
    finally:
        ops.GEMM_SCHEME = keep


def test_cfg3_default_width_model_128x256_gradients_fp64_protocol():
    """configs[3]'s per-sample work inside the full-width model (latent 1024, 768 velocity planes, every block of the
    default configuration): forward AND every parameter gradient at 128x256, B=1 - the ring-window advection scatter,
    the GlobalBias projection adjoint behind the GEMM epilogue, split-k weight gradients over 32,768 points - on the
    DEFAULT arithmetic (bf16x3), two ADR layers (the second layer's advection differentiates through the first's),
    by the fp64 protocol of SURVEY 8c(iii): the HIP gradient's distance to the fp64 oracle against the CPU-fp32
    oracle's own, as a DISTRIBUTION over three (input, cotangent) seeds.

    Why a distribution (round 4, tools/cfg3_grad_dist.py, profiles/r04_cfg3_grad_dist.json: 8 seeds): the chain
    amplifies rounding ~1e5-fold (CPU-fp32 itself up to 1e-1 off fp64 on single parameters), every parameter's error
    of one seed is the same few amplified roundings, so the per-seed median of the norm-wise ratio GPU / CPU-fp32
    moves between 1.3 and 3.3 for EVERY arithmetic:
        CPU-fp32 with permuted GEMM summation order (control)   median of medians 1.01   (0.60 .. 1.36)
        f32-MFMA GEMMs (exact)                                    2.23   (1.26 .. 2.35), one seed of eight fails the
                                                                         single-seed bounds below (11 parameters)
        bf16x3 (default, sign checkerboard)                       2.04   (1.28 .. 3.25), one seed of eight (42)
        bf16x3 of rounds 1-3 (no checkerboard)                    2.02   (1.17 .. 4.09), and 24-25 x on
                                                                         reaction.1.0-ChannelNorm.bias in EVERY seed
    - the GPU GEMMs carry 8-10 u of zero-mean noise where oneDNN's blocked sums carry 3 (tools/gemm_bias_check.py), hence
    ~2 for both GPU arithmetics; what set the old bf16x3 apart was not its noise but an OFFSET shared by all outputs
    (the bf16 MFMA floors its accumulator when a k-tile's product sum outweighs it), which sums over pixels added up
    coherently.  The checkerboard of negated-space blocks (csrc/gemm.hip) removes it; this test would catch its return.

    Asserted: forward <= 1e-5 per seed; median over the seeds of the per-seed median ratio <= 3 and every per-seed
    median <= 4.5; at least two of the three seeds pass the single-seed bounds of tests/test_hip_model.py for EVERY
    parameter (8 x max-abs, 5 x norm-wise); no parameter of any seed beyond 12 x norm-wise and the pixel-sum gradient
    that exposed the offset (reaction.1.0-ChannelNorm.bias) within 6 x in every seed.  The third seed (200 s of host
    fp64) only runs when the first two do not already decide the two-of-three and median-of-three criteria."""
    from tests.test_hip_model import _check_grads_by_fp64_protocol
    cfg = default_config()
    cfg.model.num_layers = 2
    H, W = 128, 256
    _, lg, og = make_grid(H, W, False)
    spec = _spec(cfg, H, W)
    medians, passed, worst_all, sentinel = [], 0, 0.0, []
    for seed in range(3):
        model = _build(cfg, lg, og, bias_scale=0.05)
        x = _smooth(seeded(23 + 10 * seed, 1, 186, H, W)) * 4.0
        x[:, -2], x[:, -1] = lg, og
        ct = _smooth(seeded(24 + 10 * seed, 1, 97, H, W)) * 4.0
        y32, g32 = _oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float32)
        y64, g64 = _oracle_grads_ckpt(model, spec, x, ct, lg, og, torch.float64)
        got = model(x.cuda())
        (got * ct.cuda()).sum().backward()
        e, e_cpu = max_rel(got.detach().cpu(), y32), max_rel(y32, y64)
        assert e <= 1e-5, (seed, e)
        ratios = {n: rms_rel(p.grad.cpu().double(), g64[n]) / max(rms_rel(g32[n].double(), g64[n]), 1e-12)
                  for n, p in model.named_parameters() if n in g64 and float(g64[n].abs().max()) > 0}
        rs = sorted(ratios.values())
        medians.append(rs[len(rs) // 2])
        worst_all = max(worst_all, rs[-1])
        sentinel.append(ratios["reaction.1.0-ChannelNorm.bias"])
        try:
            _check_grads_by_fp64_protocol(model, g32, g64, factor=8.0, rms_factor=5.0)
            ok = True
        except AssertionError as exc:
            ok = False
            print("seed %d: outside the single-seed bounds: %d parameters" % (seed, len(exc.args[0])))
        passed += ok
        print("cfg3 128x256 L=2 bf16x3 seed %d: forward %.2e (cpu32 vs fp64 %.2e); norm-wise ratio gpu / cpu32 median %.2f "
              "max %.2f; ChannelNorm.bias %.2f; single-seed bounds %s" % (seed, e, e_cpu, medians[-1], rs[-1], sentinel[-1], ok))
        del model, got
        torch.cuda.empty_cache()
        if seed == 1 and passed == 2 and max(medians) <= 3.0:
            break       # two of three pass and the median of three is <= 3 whatever the third seed gives
    med = sorted(medians)[1] if len(medians) == 3 else max(medians)
    assert med <= 3.0 and max(medians) <= 4.5, medians
    assert passed >= 2, (passed, medians)
    assert worst_all <= 12.0, worst_all
    assert max(sentinel) <= 6.0, sentinel


def test_cfg3_train_step_b8_properties():
    """configs[3] per GPU: one training step (forward + ParadisLoss + backward + AdamW) of the default model at
    128x256 with the per-GPU batch of 8.  Too large for the CPU oracle, so size-independent properties:
    everything finite, the batch loss equals the mean of the eight per-sample losses (samples never interact:
    reference trainer.py:508-576), the batch gradient equals the mean of per-sample gradients on a probe
    parameter set, the step fits one MI355X and every parameter moves."""
    from paradis_model_amd.harness import TrainStep, rollout_loss
    from paradis_model_amd.loss import build_loss
    cfg = default_config()
    H, W, B = 128, 256, 8
    lat_deg, lg, og = make_grid(H, W, False)
    model = _build(cfg, lg, og)
    loss_fn = build_loss(cfg, lat_deg).cuda()
    batch = tuple(t.cuda() for t in _batch(B, 1, H, W, lg, og, seed0=300))
    probes = [n for n, _ in model.named_parameters()
              if n in ("alpha_adv", "velocity_nets.3.1-SepConv.pointwise.weight", "advection.5.up_projection.0-CLinear.conv.bias",
                       "reaction.7.0-ChannelNorm.weight", "diffusion.0.0-GlobalBias.U", "output_proj.2-CLinear.conv.weight")]
    assert len(probes) == 6, probes
    named = dict(model.named_parameters())
    # per-sample losses and gradients of the probes
    per_loss, per_grad = [], {n: 0.0 for n in probes}
    for b in range(B):
        model.zero_grad(set_to_none=True)
        one = tuple(t[b:b + 1] for t in batch)
        lb, _ = rollout_loss(model, loss_fn, one, num_common=83, n_inputs=2)
        per_loss.append(float(lb))
        for n in probes:
            per_grad[n] = per_grad[n] + named[n].grad.detach().double() / B
    before = {n: p.detach().clone() for n, p in named.items()}
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    step = TrainStep(model, loss_fn, cfg, num_common=83, n_inputs=2)
    total = step(batch)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    print("cfg3 B=8 128x256 train step: peak HBM %.1f GB, loss %.6f" % (peak / 1e9, float(total)))
    assert peak < 288e9 and torch.isfinite(total)
    mean = sum(per_loss) / B
    assert abs(mean - float(total)) <= 2e-6 * abs(mean), (mean, float(total))
    for n in probes:
        e = max_rel(named[n].grad.double(), per_grad[n])
        assert e <= 2e-5, (n, e)           # fp32 sums in a different order (8 samples at once vs one by one)
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in named.items())
    assert moved == len(named), (moved, len(named))
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    del model, batch, loss_fn, step
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# configs[4]: 721x1440 with pole rows
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["bicubic", "bilinear"])
def test_cfg4_advect_721x1440_vs_oracle(mode):
    """sl_advect forward on the 0.25 degree grid (pole rows, tiled schedule) against the oracle, K=2
    planes, by the SURVEY 8c(iii) protocol (error vs fp64 against the CPU-fp32 oracle's own)."""
    from paradis_model_amd import ops
    H, W, B, K = 721, 1440, 1, 2
    _, lg, og = make_grid(H, W, True)
    f = seeded(31, B, K, H, W)
    # smooth field + velocities of a few cells: what the model produces (white noise would make the
    # fp32 coordinate rounding, 1e-4 cells at 1440 columns, dominate any comparison)
    k5 = torch.ones(1, 1, 9, 9) / 81.0
    f = torch.nn.functional.conv2d(f.reshape(-1, 1, H, W), k5, padding=4).reshape(B, K, H, W)
    u = seeded(32, B, K, H, W, scale=0.05)
    v = seeded(33, B, K, H, W, scale=0.05)
    dt = 0.196887
    want32 = O.sl_advect_core(f, u, v, dt, O.GridGeometry(lg, og), mode)
    want64 = O.sl_advect_core(f.double(), u.double(), v.double(), dt,
                              O.GridGeometry(lg.double(), og.double()), mode)
    got = ops.sl_advect(f.cuda(), u.cuda(), v.cuda(), ops.AdvectGeometry(lg, og), dt, mode).cpu()
    e_cpu, e_gpu, r32 = rms_rel(want32, want64), rms_rel(got, want64), rms_rel(got, want32)
    print("cfg4 advect %s: rms vs cpu32 %.2e, vs fp64 gpu %.2e cpu %.2e" % (mode, r32, e_gpu, e_cpu))
    assert e_gpu <= 1.5 * e_cpu + 2e-7, (e_gpu, e_cpu)
    assert r32 <= 1.5 * e_cpu + 2e-7, (r32, e_cpu)      # measured 0.77 x (round 3 asserted 3 x)
    # pole rows carry their longitudinal mean
    assert float((got[..., 0, :] - got[..., 0, :1]).abs().max()) == 0.0
    assert float((got[..., -1, :] - got[..., -1, :1]).abs().max()) == 0.0


def test_cfg4_reduced_model_721x1440_forward_vs_oracle():
    """Reduced model on the 0.25 degree grid against the oracle.  At 1440 columns one ulp of a sample
    coordinate is 1.2e-4 cells, so two fp32 implementations of the operator differ by more than 1e-5
    (SURVEY.md section 0 item 7): the HIP result is judged against the fp64 oracle next to the CPU-fp32
    oracle's own distance to it (SURVEY 8c iii), and must stay within that distance of the fp32 oracle."""
    cfg = reduced_config()
    H, W = 721, 1440
    _, lg, og = make_grid(H, W, True)
    model = _build(cfg, lg, og)
    x = seeded(41, 1, 186, H, W)
    x[:, -2], x[:, -1] = lg, og
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    p64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in params.items()}
    with torch.no_grad():
        want = O.paradis_forward(params, _spec(cfg, H, W), x, lg, og, interp_impl="aten_ref")
        want64 = O.paradis_forward(p64, _spec(cfg, H, W), x.double(), lg.double(), og.double())
        got = model(x.cuda()).cpu()
    e_cpu, e_gpu, e32 = max_rel(want, want64), max_rel(got, want64), max_rel(got, want)
    r_cpu, r_gpu = rms_rel(want, want64), rms_rel(got, want64)
    print("cfg4 721x1440 reduced-model forward: max-rel vs fp64 gpu %.2e cpu %.2e (rms %.2e / %.2e); vs cpu32 %.2e"
          % (e_gpu, e_cpu, r_gpu, r_cpu, e32))
    assert r_gpu <= 1.5 * r_cpu + 2e-7, (r_gpu, r_cpu)
    assert e_gpu <= 2.0 * e_cpu + 1e-6, (e_gpu, e_cpu)
    assert e32 <= 3.0 * e_cpu + 1e-6, (e32, e_cpu)


def _roll_lon(model, x, s):
    """the same model and input with every longitude-indexed quantity rolled by s columns"""
    import copy
    m2 = copy.deepcopy(model)
    with torch.no_grad():
        for n, p in m2.named_parameters():
            if n.endswith(".V"):           # GlobalBias V[rank, W]
                p.copy_(torch.roll(p, s, dims=1))
    return m2, torch.roll(x, s, dims=-1)


@pytest.mark.parametrize("H,W,poles", [(721, 1440, True)])
def test_cfg4_default_model_full_forward_properties(H, W, poles):
    """configs[4] at full size (default model, B=1): finite, reproducible bit for bit, and
    longitude-roll equivariant (see the module docstring).  The equivariance holds up to the fp32
    rounding of the sample coordinates (one ulp of a longitude index at 1440 columns is 1.2e-4 cells,
    and the wrap point moves with the roll), which eight layers carry into the output in proportion to
    the field slopes: smooth inputs, and a bound two orders above the 1e-5 of the oracle comparisons."""
    cfg = default_config()
    _, lg, og = make_grid(H, W, poles)
    model = _build(cfg, lg, og, bias_scale=0.05)
    xd = _smooth(seeded(51, 1, 186, H, W).cuda()) * 4.0
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        y1 = model(xd)
        y2 = model(xd)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated()
        assert bool(torch.isfinite(y1).all())
        assert torch.equal(y1, y2), "forward is not reproducible"
        s = W // 3
        m2, xr = _roll_lon(model, xd, s)
        yr = m2(xr)
    e = max_rel(yr, torch.roll(y1, s, dims=-1))
    r = rms_rel(yr, torch.roll(y1, s, dims=-1))
    print("cfg4 full forward: peak HBM %.1f GB, roll-equivariance max-rel %.2e rms-rel %.2e" % (peak / 1e9, e, r))
    assert peak < 288e9
    assert r <= 1e-4 and e <= 2e-3, (e, r)


def test_roll_equivariance_reduced_model_32x64():
    """The same property where rounding is small (64 columns): max-rel <= 1e-5."""
    cfg = reduced_config()
    H, W = 32, 64
    _, lg, og = make_grid(H, W, False)
    model = _build(cfg, lg, og)
    xd = seeded(61, 2, 186, H, W).cuda()
    with torch.no_grad():
        y1 = model(xd)
        m2, xr = _roll_lon(model, xd, 21)
        yr = m2(xr)
    e = max_rel(yr, torch.roll(y1, 21, dims=-1))
    print("roll-equivariance 32x64 reduced model max-rel %.2e" % e)
    assert e <= 1e-5, e
