"""The split pointwise GEMMs (fp32 in / fp32 accumulate / fp32 out on the bf16 / f16 matrix pipe) against
the exact f32-MFMA kernels and an fp64 evaluation of the same product.

Claim under test (include/paradis_hip.h, a6): the split paths are fp32 GEMMs, not reduced-precision ones:
their error against fp64 is not above the exact f32 chain's.  Bounds written here, for "bf16x3" (exact
three-term decomposition) and for "f16x2" (two f16 terms of the per-tensor scaled operands, the default):
  * max |split - fp64| / max |fp64|  <=  1.25 x the same figure of the exact kernel + 1e-7, and
  * <= 2e-6 absolutely (an fp32 dot product of length <= 1024 with |x|,|w| ~ 1 sits at ~3e-7).
What f16x2 gives up - relative accuracy of elements far below their TENSOR's largest magnitude - and its
all-or-nothing treatment of non-finite operands have their own tests below."""
import pytest
import torch

from tests._util import seeded

pytestmark = pytest.mark.gpu

SHAPES = [  # B, Ci, Co, H, W
    (2, 10, 7, 12, 16),        # one ragged tile, K < 16
    (2, 186, 64, 16, 32),      # K % 16 != 0 (the input projection's 186 channels)
    (3, 128, 97, 32, 64),      # M ragged
    (1, 256, 384, 17, 32),     # N = 544: ragged n tile, still N % 16 == 0
    (2, 130, 258, 9, 20),      # N = 180: weight-gradient falls back to the exact kernel (N % 16 != 0)
    (2, 1024, 896, 32, 64),    # the default model's widest GEMM
]


@pytest.fixture(scope="module")
def ops():
    from paradis_model_amd import ops as _ops
    return _ops


SPLIT_SCHEMES = ["f16x2", "bf16x3"]


def _run(ops, scheme, x, w, b, res, ct, act):
    keep = ops.GEMM_SCHEME
    ops.GEMM_SCHEME = ops._SCHEMES[scheme]
    try:
        ds = [t.detach().clone().cuda().requires_grad_(True) for t in (x, w, b, res)]
        y = ops.pointwise(ds[0], ds[1], ds[2], None, ds[3], act)
        y.backward(ct.cuda())
        torch.cuda.synchronize()
        return [y.detach()] + [d.grad for d in ds[:3]]
    finally:
        ops.GEMM_SCHEME = keep


def _fp64(x, w, b, res, ct, act):
    ts = [t.detach().clone().cuda().double().requires_grad_(True) for t in (x, w, b, res)]
    z = torch.einsum("oc,bchw->bohw", ts[1], ts[0]) + ts[2].view(1, -1, 1, 1)
    if act == "SiLU":
        z = torch.nn.functional.silu(z)
    elif act == "GELU":
        z = torch.nn.functional.gelu(z)
    y = z + ts[3]
    y.backward(ct.cuda().double())
    return [y.detach()] + [t.grad for t in ts[:3]]


def _err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("B,Ci,Co,H,W", SHAPES)
@pytest.mark.parametrize("act", [None, "SiLU"])
@pytest.mark.parametrize("scheme", SPLIT_SCHEMES)
def test_split_not_less_accurate_than_exact_f32(ops, scheme, B, Ci, Co, H, W, act):
    x = seeded(1, B, Ci, H, W)
    w = seeded(2, Co, Ci, scale=Ci ** -0.5)
    b = seeded(3, Co, scale=0.1)
    res = seeded(5, B, Co, H, W)
    ct = seeded(6, B, Co, H, W)
    ref = _fp64(x, w, b, res, ct, act)
    exact = _run(ops, "exact", x, w, b, res, ct, act)
    split = _run(ops, scheme, x, w, b, res, ct, act)
    for name, s, e, r in zip(("y", "gx", "gw", "gb"), split, exact, ref):
        es, ee = _err(s, r), _err(e, r)
        assert es <= 1.25 * ee + 1e-7, (name, es, ee)
        assert es <= 2e-6, (name, es)


def test_split_exactness_on_bf16_representable_inputs(ops):
    """Inputs that are exactly bf16 numbers with small integer values make every partial product and
    every partial sum exact in fp32 (f16x2: the power-of-two scaling is exact too): all three paths must
    then agree bit for bit with integer arithmetic."""
    g = torch.Generator().manual_seed(7)
    B, Ci, Co, H, W = 2, 64, 48, 8, 16
    x = torch.randint(-8, 9, (B, Ci, H, W), generator=g).float()
    w = torch.randint(-8, 9, (Co, Ci), generator=g).float()
    b = torch.zeros(Co)
    res = torch.zeros(B, Co, H, W)
    ct = torch.randint(-4, 5, (B, Co, H, W), generator=g).float()
    ref = _fp64(x, w, b, res, ct, None)
    for mode in ("f16x2", "bf16x3", "exact"):
        got = _run(ops, mode, x, w, b, res, ct, None)
        for name, a, r in zip(("y", "gx", "gw"), got, ref):
            assert torch.equal(a.double(), r), (mode, name)


def test_split_handles_wide_dynamic_range(ops):
    """bf16x3, operands spanning 12 decades: the h/m/l terms keep fp32's exponent range (bf16 has the same
    8 exponent bits), so nothing under- or overflows in the split."""
    g = torch.Generator().manual_seed(11)
    B, Ci, Co, H, W = 1, 96, 64, 8, 16
    mag = 10.0 ** (torch.rand(B, Ci, H, W, generator=g) * 12 - 6)
    x = torch.randn(B, Ci, H, W, generator=g) * mag
    w = torch.randn(Co, Ci, generator=g) * 10.0 ** (torch.rand(Co, Ci, generator=g) * 6 - 3)
    b = torch.zeros(Co)
    res = torch.zeros(B, Co, H, W)
    ct = torch.randn(B, Co, H, W, generator=g)
    ref = _fp64(x, w, b, res, ct, None)
    exact = _run(ops, "exact", x, w, b, res, ct, None)
    split = _run(ops, "bf16x3", x, w, b, res, ct, None)
    for name, s, e, r in zip(("y", "gx", "gw"), split, exact, ref):
        # element-wise relative to the fp64 magnitude scale of each output row/column is too strict
        # for cancelling sums; compare the two paths on the same max-normalised figure
        es, ee = _err(s, r), _err(e, r)
        assert es <= 1.25 * ee + 1e-7, (name, es, ee)


def test_f16x2_error_model(ops):
    """What the two-term scheme promises (include/paradis_hip.h): every operand element is represented
    to 2^-22 of itself or 2^-39 of its tensor's largest magnitude, whichever is larger.
      * tensors of any overall magnitude (1e-30 .. 1e+30): fp32-level result (the scale is per tensor);
      * a tensor whose samples differ by 2^12 in magnitude: the small samples' outputs still carry
        <= 2e-6 relative error (2^12 x 2^12 below the maxima on both operands is inside the 2^39 window);
      * entries 2^30 below the tensor's maximum: their contribution is right to 2^-39 of the maximum, i.e.
        the ABSOLUTE error of the output stays at fp32 level although those entries alone are not."""
    g = torch.Generator().manual_seed(21)
    B, Ci, Co, H, W = 2, 96, 64, 8, 16
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, generator=g) * Ci ** -0.5
    z = torch.zeros
    for sx, sw in ((1e30, 1e-30), (1e-30, 1e30), (1e-20, 1e-10), (1e15, 1e15)):
        ref = _fp64(x * sx, w * sw, z(Co), z(B, Co, H, W), torch.randn(B, Co, H, W, generator=g), None)
        got = _run(ops, "f16x2", x * sx, w * sw, z(Co), z(B, Co, H, W), torch.randn(B, Co, H, W, generator=g), None)
        assert _err(got[0], ref[0]) <= 2e-6, (sx, sw, _err(got[0], ref[0]))
    xs = x.clone()
    xs[1] *= 2.0 ** -12                       # second sample 2^12 below the first
    ct = torch.randn(B, Co, H, W, generator=g)
    ref = _fp64(xs, w, z(Co), z(B, Co, H, W), ct, None)
    got = _run(ops, "f16x2", xs, w, z(Co), z(B, Co, H, W), ct, None)
    assert _err(got[0][1], ref[0][1]) <= 2e-6 and _err(got[0][0], ref[0][0]) <= 2e-6
    assert _err(got[1][1], ref[1][1]) <= 2e-6       # gx of the small sample
    xt = x.clone()
    xt[:, ::2] *= 2.0 ** -30                  # every other channel 2^30 below the rest
    ref = _fp64(xt, w, z(Co), z(B, Co, H, W), ct, None)
    got = _run(ops, "f16x2", xt, w, z(Co), z(B, Co, H, W), ct, None)
    assert _err(got[0], ref[0]) <= 2e-6 and _err(got[2], ref[2]) <= 2e-6


def test_f16x2_non_finite_operand_poisons_the_product(ops):
    """An Inf or NaN anywhere in an operand tensor makes its largest magnitude non-finite: the whole
    product comes out NaN (loudly wrong), never silently rescaled.  bf16x3 / exact confine it to the
    outputs it touches (test_split_extreme_magnitudes)."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 64, 8, 16, generator=g)
    w = torch.randn(48, 64, generator=g) * 0.1
    for bad in (float("inf"), float("-inf"), float("nan")):
        xb = x.clone(); xb[0, 7, 1, 4] = bad
        assert torch.isnan(_fwd_only(ops, "f16x2", xb, w)).all()
        wb = w.clone(); wb[5, 9] = bad
        assert torch.isnan(_fwd_only(ops, "f16x2", x, wb)).all()
    assert torch.isfinite(_fwd_only(ops, "f16x2", x, w)).all()
    # all-zero operands: zero result, no NaN from the scale
    assert (_fwd_only(ops, "f16x2", torch.zeros_like(x), w) == 0).all()
    assert (_fwd_only(ops, "f16x2", x, torch.zeros_like(w)) == 0).all()


def test_amax_partials(ops):
    """The reduction feeding the f16x2 scale: maximum of the words = max |x|, on contiguous tensors,
    channel-sliced views, sizes that are not multiples of four, and with a NaN inside."""
    g = torch.Generator().manual_seed(9)
    for shape in ((2, 5, 7, 9), (3, 64, 32, 64), (1, 1, 1, 1)):
        x = torch.randn(*shape, generator=g).cuda()
        p = ops._amax_partials(x)
        assert p.shape == (ops.AMAX_PARTIALS,) and p.dtype == torch.int32
        got = p.max().view(1).view(torch.float32)
        assert float(got) == float(x.abs().max())
    base = torch.randn(2, 12, 8, 16, generator=g).cuda()
    base[:, 8:] *= 100.0
    view = base[:, 2:8]                       # channel slice: batch stride != C H W
    assert float(ops._amax_partials(view).max().view(1).view(torch.float32)) == float(view.abs().max())
    base[1, 3, 2, 2] = float("nan")
    assert torch.isnan(ops._amax_partials(view).max().view(1).view(torch.float32)).all()


@pytest.mark.parametrize("scheme", SPLIT_SCHEMES)
def test_split_random_ragged_shapes(ops, scheme):
    """30 random small shapes (every combination of ragged M / K / N tiles, K < 8, N < 32, N % 16 != 0 for
    which the weight gradient takes the exact kernel): all four results within 2e-6 of fp64."""
    g = torch.Generator().manual_seed(2024)
    for case in range(30):
        B = int(torch.randint(1, 4, (1,), generator=g))
        Ci = int(torch.randint(1, 300, (1,), generator=g))
        Co = int(torch.randint(1, 300, (1,), generator=g))
        H = int(torch.randint(1, 24, (1,), generator=g))
        W = int(torch.randint(1, 40, (1,), generator=g))
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, generator=g) * Ci ** -0.5
        b = torch.randn(Co, generator=g) * 0.1
        res = torch.randn(B, Co, H, W, generator=g)
        ct = torch.randn(B, Co, H, W, generator=g)
        ref = _fp64(x, w, b, res, ct, "SiLU")
        got = _run(ops, scheme, x, w, b, res, ct, "SiLU")
        for name, a, r in zip(("y", "gx", "gw", "gb"), got, ref):
            assert torch.isfinite(a).all(), (case, name)
            assert _err(a, r) <= 2e-6, (case, (B, Ci, Co, H, W), name, _err(a, r))


def test_batched_gemm_split_vs_exact():
    """paradis_bgemm (Newton-Schulz products of Muon) in both arithmetics against fp64."""
    from paradis_model_amd._lib import dptr, lib, stream_ptr
    g = torch.Generator().manual_seed(5)
    for (T, M, K, N) in [(3, 96, 160, 96), (2, 130, 130, 258), (5, 64, 64, 64), (1, 896, 1152, 896)]:
        A = (torch.randn(T, M, K, generator=g) * K ** -0.5).cuda()
        Bm = torch.randn(T, K, N, generator=g).cuda()
        ref = A.double() @ Bm.double()
        errs = {}
        for split in (False, True):
            C = torch.empty(T, M, N, device="cuda")
            ws = torch.empty(T * lib.paradis_pw_gemm_split_bytes(M, K, 3), dtype=torch.uint8, device="cuda") if split else None
            rc = lib.paradis_bgemm(dptr(A), None, dptr(Bm), dptr(C), T, M, K, N, M * K, 0, K * N, M * N,
                                   dptr(ws), stream_ptr())
            assert rc == 0
            torch.cuda.synchronize()
            errs[split] = _err(C, ref)
        assert errs[True] <= 1.25 * errs[False] + 1e-7 and errs[True] <= 2e-6, ((T, M, K, N), errs)


def _fwd_only(ops, scheme, x, w):
    if isinstance(scheme, bool):
        scheme = "bf16x3" if scheme else "exact"
    keep = ops.GEMM_SCHEME
    ops.GEMM_SCHEME = ops._SCHEMES[scheme]
    try:
        with torch.no_grad():
            return ops.pointwise(x.cuda(), w.cuda()).cpu()
    finally:
        ops.GEMM_SCHEME = keep


def test_split_extreme_magnitudes(ops):
    """Edge semantics of the bf16 h/m/l split next to the exact f32 kernel (documented in DESIGN.md 4.1b):
      * operands of magnitude 1e+-30 whose products are O(1): fp32-level result (bf16 has fp32's exponent
        range, nothing over- or underflows in the split);
      * operands below ~1e-33: the m / l terms (2^-8, 2^-16 of the operand) fall below fp32's smallest
        normal number and are flushed - the split degrades gracefully towards bf16 precision of THAT
        operand (<= 2^-15 at 1e-35, <= 2^-7 at 1e-37..1e-38), where the exact kernel keeps fp32 precision.
        Activations and gradients of this model sit 20+ decades above that; PARADIS_GEMM=exact is the
        answer for data that does not;
      * |x| within half a bf16 ulp of FLT_MAX (>= 3.39e38) rounds to +-inf in the leading term: such an
        operand gives NaN where the exact kernel may still be finite;
      * Inf / NaN operands: every output they touch is non-finite in both arithmetics (the split turns
        Inf into NaN: Inf - Inf in the residual), every other output is untouched."""
    g = torch.Generator().manual_seed(77)
    B, Ci, Co, H, W = 1, 64, 48, 8, 16
    base_x = torch.randn(B, Ci, H, W, generator=g)
    base_w = torch.randn(Co, Ci, generator=g) * Ci ** -0.5
    ref = torch.einsum("oc,bchw->bohw", base_w.double(), base_x.double())
    for sx, sw in ((1e30, 1e-30), (1e-30, 1e30)):
        x, w = base_x * sx, base_w * sw
        want = torch.einsum("oc,bchw->bohw", w.double(), x.double())
        ys, ye = _fwd_only(ops, True, x, w), _fwd_only(ops, False, x, w)
        assert torch.isfinite(ys).all()
        es, ee = _err(ys, want), _err(ye, want)
        assert es <= 1.25 * ee + 1e-7 and es <= 2e-6, (sx, es, ee)
    # subnormal low terms of a tiny operand: graceful loss of its low bits, never garbage
    for tiny, bound in ((1e-35, 2.0 ** -15), (1e-37, 2.0 ** -7)):
        for x, w in ((base_x * tiny, base_w / tiny), (base_x / tiny * 1e-2, base_w * tiny * 1e2)):
            want = torch.einsum("oc,bchw->bohw", w.double(), x.double())
            ys = _fwd_only(ops, True, x, w)
            assert torch.isfinite(ys).all() and _err(ys, want) <= bound, (tiny, _err(ys, want))
    # the top half-ulp of the fp32 range
    x = base_x.clone()
    x[0, 3, 2, 5] = 3.0e38
    w = base_w * 1e-38
    ys, ye = _fwd_only(ops, True, x, w), _fwd_only(ops, False, x, w)
    assert torch.isfinite(ys).all() and torch.isfinite(ye).all()
    x[0, 3, 2, 5] = 3.4e38              # rounds to +inf as a bf16 leading term
    ys, ye = _fwd_only(ops, True, x, w), _fwd_only(ops, False, x, w)
    assert torch.isfinite(ye).all() and not torch.isfinite(ys[0, :, 2, 5]).any()
    # Inf and NaN operands: same set of non-finite outputs in both arithmetics
    for bad in (float("inf"), float("-inf"), float("nan")):
        x = base_x.clone()
        x[0, 7, 1, 4] = bad
        ys, ye = _fwd_only(ops, True, x, base_w), _fwd_only(ops, False, x, base_w)
        assert torch.equal(torch.isfinite(ys), torch.isfinite(ye)), bad
        assert not torch.isfinite(ys[0, :, 1, 4]).any()
        keep = torch.isfinite(ys)
        assert float((ys[keep].double() - ref[keep]).abs().max() / ref.abs().max()) <= 2e-6
        w = base_w.clone()
        w[5, 9] = bad
        ys, ye = _fwd_only(ops, True, base_x, w), _fwd_only(ops, False, base_x, w)
        assert torch.equal(torch.isfinite(ys), torch.isfinite(ye)), bad
        assert not torch.isfinite(ys[0, 5]).any() and torch.isfinite(ys[0, :5]).all()


def test_amax_side_outputs(ops):
    """Producer kernels leave the partial maxima of what they store attached to their output (f16x2
    scheme): equal to a read pass (an upper bound for the advection's pole rows), picked up by
    ops.amax_partials without a launch, and dropped when the tensor is modified in place."""
    from tests._util import make_grid
    keep, keep_traced = ops.GEMM_SCHEME, ops.TRACED
    ops.GEMM_SCHEME = ops.GEMM_F16X2
    ops.TRACED = False          # (tracing any op - torch.compile, opcheck in other tests - switches the channel off)
    try:
        g = torch.Generator().manual_seed(31)
        B, C, H, W = 2, 64, 32, 64

        def amax_of(words):
            return float(words.max().view(1).view(torch.float32))

        def check(t, exact=True):
            words = ops._amax_lookup(t)
            assert words is not None, "no side output attached"
            got, want = amax_of(words), float(t.detach().abs().max())
            assert got == want if exact else (want <= got <= 4 * want), (got, want)
            assert ops._amax_partials(t) is words      # no read pass
            return words

        x = torch.randn(B, C, H, W, generator=g).cuda()
        w = (torch.randn(C, generator=g) + 1).cuda()
        b = torch.randn(C, generator=g).cuda()
        check(ops.channel_norm(x, w, b, 1e-5))
        dw = torch.randn(C, 1, 5, 5, generator=g).cuda()
        check(ops.dwconv_geo(x, dw, b))
        pw = (torch.randn(48, C, generator=g) * 0.1).cuda().requires_grad_(True)
        xin = x.clone().requires_grad_(True)
        y, zpre = ops.pointwise(xin, pw, None, None, None, "SiLU", defer_act_grad=True)   # chained CLinear layers
        check(y)
        assert ops._amax_lookup(ops.pointwise(xin, pw, None, None, None, "SiLU")) is None   # (only then)
        _, lg, og = make_grid(H, W, False)
        geom = ops.AdvectGeometry(lg, og)
        vel = (torch.randn(B, 2 * C, H, W, generator=g) * 0.3).cuda()
        check(ops.sl_advect_vel(x, vel, geom, 0.05, "bicubic"), exact=False)
        # backward producers: act_backward (dz of the GEMM above) and the dgrad epilogue
        gy = torch.randn(B, 48, H, W, generator=g).cuda()
        dz = ops._act_backward(gy, y.detach(), 1)
        check(dz)
        gx = ops._pw_gemm_dgrad(dz, pw.detach(), x, 1, None)     # deferred chain: gx = W^T dz * act'(x_pre)
        check(gx)
        # cotangent producers: ChannelNorm backward (gx1), gated blend backward (gadv), advection backward
        mean = torch.zeros(B, H * W).cuda(); rstd = torch.ones(B, H * W).cuda()
        gx1 = ops._channel_norm_backward(x, x, None, w, mean, rstd, None)[0]
        check(gx1)
        alpha = torch.randn(C, generator=g).cuda()
        check(ops._gated_blend_backward(x, x, x * 0.5, alpha)[1])
        gf, gvel = ops._sl_advect_vel_backward(x, x, vel, *ops._geom_args(geom, x.device, 0.05, "bicubic", None))
        check(gf)
        check(gvel)
        # an in-place update invalidates the attachment
        gx.mul_(2.0)
        assert ops._amax_lookup(gx) is None
        assert amax_of(ops._amax_partials(gx)) == float(gx.abs().max())
        # ragged GEMM tiles (guarded epilogue path) and a shape smaller than one tile
        for (Co, Ci, h, wd) in ((97, 186, 9, 20), (7, 10, 4, 8)):
            xs = torch.randn(1, Ci, h, wd, generator=g).cuda()
            ws = torch.randn(Co, Ci, generator=g).cuda()
            check(ops.pointwise(xs, ws, None, None, None, "SiLU", defer_act_grad=True)[0])
        # once an op has been traced with fake tensors the channel stays off: a compiled graph may update a
        # custom op's output in place without touching its version counter
        ops.TRACED = True
        assert ops._amax_lookup(ops.dwconv_geo(x, dw, b)) is None
    finally:
        ops.GEMM_SCHEME, ops.TRACED = keep, keep_traced


def test_model_gradients_do_not_depend_on_the_amax_side_channel(ops):
    """A training step of the reduced model with the producers' amax side outputs and with one read pass per
    GEMM operand: same loss and gradients (the scales are powers of two; only the advection's upper bound can
    move one by a binade)."""
    from paradis_model_amd.config import reduced_config, stub_datamodule
    from paradis_model_amd.model import Paradis
    from tests._util import make_grid, max_rel
    keep = (ops.GEMM_SCHEME, ops.TRACED, ops.AMAX_SIDE_OUTPUTS)
    ops.GEMM_SCHEME, ops.TRACED = ops.GEMM_F16X2, False
    try:
        cfg = reduced_config()
        _, lg, og = make_grid(32, 64, False)
        torch.manual_seed(42)
        model = Paradis(stub_datamodule(cfg), cfg, lg, og).cuda()
        x = seeded(3, 2, 186, 32, 64).cuda()
        res = []
        for side in (True, False):
            ops.AMAX_SIDE_OUTPUTS = side
            model.zero_grad(set_to_none=True)
            xd = x.clone().requires_grad_(True)
            y = model(xd)
            y.square().mean().backward()
            res.append((y.detach(), xd.grad, torch.cat([p.grad.flatten() for p in model.parameters()])))
        for a, b in zip(res[0], res[1]):
            assert max_rel(a, b) <= 1e-5
    finally:
        ops.GEMM_SCHEME, ops.TRACED, ops.AMAX_SIDE_OUTPUTS = keep
