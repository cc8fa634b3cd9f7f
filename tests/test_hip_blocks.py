"""GPU parity of every block-level HIP op against the CPU oracle and the reference goldens (G3).
Tolerance (SURVEY.md 8c ii): max-abs(diff)/max-abs(ref) <= 1e-5 forward; gradients <= 1e-4."""
import pytest
import torch

from oracle import paradis_oracle as O
from tests._util import load_golden, max_rel, seeded

pytestmark = pytest.mark.gpu
FWD, BWD = 1e-5, 1e-4


@pytest.fixture(scope="module")
def ops():
    from paradis_model_amd import ops as _ops
    return _ops


def _dev(t):
    return t.detach().clone().cuda().requires_grad_(True)


def _cmp(got, want, tol, what):
    e = max_rel(got.detach().cpu(), want.detach())
    assert e <= tol, (what, e)
    return e


# ----------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 10, 7, 12, 16), (2, 186, 64, 16, 32), (3, 128, 97, 32, 64),
                                         (1, 256, 384, 17, 32), (2, 130, 258, 32, 64)])
@pytest.mark.parametrize("act", [None, "SiLU", "GELU"])
def test_pointwise_gemm(ops, B, Ci, Co, H, W, act):
    x = seeded(1, B, Ci, H, W)
    w = seeded(2, Co, Ci, 1, 1, scale=Ci ** -0.5)
    b = seeded(3, Co, scale=0.1)
    bm = seeded(4, Co, H, W, scale=0.2)
    res = seeded(5, B, Co, H, W)
    ct = seeded(6, B, Co, H, W)
    ts = [t.clone().requires_grad_(True) for t in (x, w, b, bm, res)]
    y = O.pointwise(ts[0], ts[1], ts[2]) + ts[3].unsqueeze(0)
    if act:
        y = O.activation(y, act)
    y = y + ts[4]
    y.backward(ct)
    ds = [_dev(t) for t in (x, w, b, bm, res)]
    yd = ops.pointwise(ds[0], ds[1], ds[2], ds[3], ds[4], act)
    yd.backward(ct.cuda())
    _cmp(yd, y, FWD, "y")
    for name, d, t in zip(("gx", "gw", "gb", "gmap", "gres"), ds, ts):
        _cmp(d.grad, t.grad, BWD, name)


@pytest.mark.parametrize("B,Ci,Co,H,W,cin,R", [(2, 10, 8, 12, 16, 4, 6), (2, 128, 160, 32, 64, 8, 16),
                                               (1, 130, 260, 17, 32, 3, 5),
                                               (1, 32, 64, 64, 128, 8, 6),     # P >= 8192: the row-wise gPw kernel, two rows per workgroup
                                               (1, 16, 1028, 64, 128, 8, 6),   # ... four rows per workgroup
                                               (1, 8, 16, 40, 300, 5, 7)])     # W > 256, odd channel / rank counts
@pytest.mark.parametrize("act", [None, "SiLU"])
def test_pointwise_fused_global_bias_projection(ops, B, Ci, Co, H, W, cin, R, act):
    """GlobalBias with projection applied inside the GEMM epilogue (no [Co,H,W] map in the forward)
    == reference GlobalBias(x) = x + P(einsum(A,U,V)) (model/blocks.py:190-196) after a 1x1 conv."""
    x = seeded(1, B, Ci, H, W)
    w = seeded(2, Co, Ci, 1, 1, scale=Ci ** -0.5)
    b = seeded(3, Co, scale=0.1)
    A, U, V = seeded(4, cin, R, scale=0.5), seeded(5, R, H), seeded(6, R, W)
    Pw = seeded(7, Co, cin, scale=0.5)
    res = seeded(8, B, Co, H, W)
    ct = seeded(9, B, Co, H, W)
    ts = [t.clone().requires_grad_(True) for t in (x, w, b, A, U, V, Pw, res)]
    bm = O.global_bias_map(ts[3], ts[4], ts[5], ts[6])
    y = O.pointwise(ts[0], ts[1], ts[2]) + bm.unsqueeze(0)
    if act:
        y = O.activation(y, act)
    y = y + ts[7]
    y.backward(ct)
    ds = [_dev(t) for t in (x, w, b, A, U, V, Pw, res)]
    m8 = ops.global_bias_m8(ds[3], ds[4], ds[5])
    yd = ops.pointwise(ds[0], ds[1], ds[2], None, ds[7], act, bias_proj=(m8, ds[6]))
    yd.backward(ct.cuda())
    _cmp(yd, y, FWD, "y")
    for name, d, t in zip(("gx", "gw", "gb", "gA", "gU", "gV", "gPw", "gres"), ds, ts):
        _cmp(d.grad, t.grad, BWD, name)


def test_pointwise_channel_slice_input(ops):
    big = seeded(1, 2, 40, 16, 32)
    w = seeded(2, 24, 16, 1, 1, scale=0.25)
    want = O.pointwise(big[:, 8:24], w, None)
    got = ops.pointwise(big.cuda()[:, 8:24], w.cuda())
    _cmp(got, want, FWD, "slice")


def test_clinear_golden(ops):
    rec = load_golden("g3_blocks.pt")["clinear"]
    x, w, b = _dev(rec["x"]), _dev(rec["params"]["conv.weight"]), _dev(rec["params"]["conv.bias"])
    y = ops.pointwise(x, w, b)
    y.backward(rec["cot"].cuda())
    _cmp(y, rec["y"], FWD, "y")
    _cmp(x.grad, rec["gx"], BWD, "gx")
    _cmp(w.grad, rec["grads"]["conv.weight"], BWD, "gw")
    _cmp(b.grad, rec["grads"]["conv.bias"], BWD, "gb")


@pytest.mark.parametrize("M,K", [(7, 10), (130, 258), (1024, 1068), (97, 512)])
def test_weight_images_pair_launch_equals_the_two_single_launches(ops, M, K):
    """paradis_pw_gemm_split_weights_pair: the bf16x3 images of W and W^T from one launch are byte-identical to the
    two single launches (and so are the bf16-mixed scheme's, paradis_pw_gemm_split_weights_pair_scheme); the recorded
    forward of ``pointwise`` caches both (one launch), a no-grad forward only W's."""
    from paradis_model_amd._lib import lib
    w = seeded(7, M, K).cuda()
    images_t = {}
    for sch in (ops.GEMM_BF16, ops.GEMM_BF16X3):
        n, nt = lib.paradis_pw_gemm_split_bytes(M, K, sch), lib.paradis_pw_gemm_split_bytes(K, M, sch)
        a, at = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(nt, dtype=torch.uint8, device="cuda")
        b, bt = torch.zeros_like(a), torch.zeros_like(at)
        assert lib.paradis_pw_gemm_split_weights(ops.dptr(w), M, K, 0, sch, ops.dptr(a), ops.stream_ptr()) == 0
        assert lib.paradis_pw_gemm_split_weights(ops.dptr(w), M, K, 1, sch, ops.dptr(at), ops.stream_ptr()) == 0
        assert lib.paradis_pw_gemm_split_weights_pair_scheme(ops.dptr(w), M, K, sch, ops.dptr(b), ops.dptr(bt), ops.stream_ptr()) == 0
        assert torch.equal(a, b) and torch.equal(at, bt), sch
        images_t[sch] = at
    assert lib.paradis_pw_gemm_split_weights_pair_scheme(ops.dptr(w), M, K, ops.GEMM_F16X2, ops.dptr(b), ops.dptr(bt), ops.stream_ptr()) == 1
    # the bf16-mixed scheme's recorded forward hands its backward the W^T image of the same launch as well
    wq = torch.nn.Parameter(w.clone().reshape(M, K, 1, 1))
    xq = seeded(8, 2, K, 8, 16).cuda().requires_grad_(True)
    yq = ops.pointwise(xq, wq, scheme=ops.GEMM_BF16)
    assert yq.grad_fn.wt_image is not None and torch.equal(yq.grad_fn.wt_image, images_t[ops.GEMM_BF16])
    yq.sum().backward()
    n, nt = lib.paradis_pw_gemm_split_bytes(M, K, ops.GEMM_BF16X3), lib.paradis_pw_gemm_split_bytes(K, M, ops.GEMM_BF16X3)
    a, at = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(nt, dtype=torch.uint8, device="cuda")
    b, bt = torch.zeros_like(a), torch.zeros_like(at)
    st = ops.stream_ptr()
    assert lib.paradis_pw_gemm_split_weights(ops.dptr(w), M, K, 0, ops.GEMM_BF16X3, ops.dptr(a), st) == 0
    assert lib.paradis_pw_gemm_split_weights(ops.dptr(w), M, K, 1, ops.GEMM_BF16X3, ops.dptr(at), st) == 0
    assert lib.paradis_pw_gemm_split_weights_pair(ops.dptr(w), M, K, ops.dptr(b), ops.dptr(bt), st) == 0
    assert torch.equal(a, b) and torch.equal(at, bt)
    assert lib.paradis_pw_gemm_split_weights_pair(ops.dptr(w), M, K, ops.dptr(b), ops.dptr(b), st) == 1
    # a recorded forward writes both images in one launch and hands W^T's to its backward through the autograd context;
    # a no-grad forward writes only W's; nothing is cached across calls (round 6: no stale-image mode)
    wp = torch.nn.Parameter(w.clone().reshape(M, K, 1, 1))
    x = seeded(8, 2, K, 8, 16).cuda().requires_grad_(True)
    with torch.no_grad():
        ops.pointwise(x, wp, scheme=ops.GEMM_BF16X3)
    assert not ops._IMAGES and ops._take_wt(wp) is None
    y = ops.pointwise(x, wp, scheme=ops.GEMM_BF16X3)
    img_t = y.grad_fn.wt_image
    assert img_t is not None and torch.equal(img_t, at)
    y.sum().backward()
    assert not ops._IMAGES
    # opt-in cache: inside frozen_weights() the images are reused across calls and dropped on leaving
    with ops.frozen_weights():
        with torch.no_grad():
            ops.pointwise(x, wp, scheme=ops.GEMM_BF16X3)
            img = ops._IMAGES[(id(wp), False, ops.GEMM_BF16X3)][4]
            ops.pointwise(x, wp, scheme=ops.GEMM_BF16X3)
            assert ops._IMAGES[(id(wp), False, ops.GEMM_BF16X3)][4] is img and torch.equal(img, a)
    assert not ops._IMAGES


# ----------------------------------------------------------------------------------- depthwise
@pytest.mark.parametrize("k", [1, 3, 5, 7, 9, 11])
@pytest.mark.parametrize("B,C,H,W", [(2, 6, 12, 16), (2, 5, 33, 64), (1, 3, 70, 130), (2, 4, 9, 8),
                                     (2, 6, 32, 64), (3, 5, 16, 64), (1, 3, 96, 200), (2, 3, 32, 68)])   # whole-plane 16-byte staging (2), staged tiles (33x64, 96x200, 32x68 at k = 5)
@pytest.mark.parametrize("bias", [False, True])
def test_dwconv_geo(ops, k, B, C, H, W, bias):
    if (k - 1) // 2 > H - 2 or k - 1 > W or (k > 7 and (H < 2 * k or W < 2 * k)):
        pytest.skip("grid too small")
    x = seeded(1, B, C, H, W)
    w = seeded(2, C, 1, k, k, scale=1.0 / k)
    b = seeded(3, C) if bias else None
    ct = seeded(4, B, C, H, W)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    y = O.depthwise_geo(xr, wr, br)
    y.backward(ct)
    xd, wd = _dev(x), _dev(w)
    bd = _dev(b) if bias else None
    yd = ops.dwconv_geo(xd, wd, bd)
    yd.backward(ct.cuda())
    _cmp(yd, y, FWD, "y")
    _cmp(xd.grad, xr.grad, BWD, "gx")
    _cmp(wd.grad, wr.grad, BWD, "gw")
    if bias:
        _cmp(bd.grad, br.grad, BWD, "gb")


@pytest.mark.parametrize("k,B,C,H,W", [(5, 2, 6, 32, 64), (5, 3, 5, 16, 64), (5, 2, 5, 33, 64), (3, 2, 6, 12, 16),
                                       (7, 1, 3, 70, 130)])
def test_dwconv_geo_skip_adds_the_other_gradient_in_the_dgrad_kernel(ops, k, B, C, H, W):
    """``dwconv_geo_skip`` = ``(dwconv_geo(x), x)``: the gradient of the second output enters the data-gradient kernel
    as an addend (paradis_dwconv_geo_dgrad_add; whole-plane and tiled kernels).  Oracle: the stencil's input consumed
    twice, as the gated blend and the advection's down-projection do (reference model/paradis.py:236-240)."""
    x, w, b = seeded(1, B, C, H, W), seeded(2, C, 1, k, k, scale=1.0 / k), seeded(3, C)
    ct, ct2 = seeded(4, B, C, H, W), seeded(5, B, C, H, W)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    (O.depthwise_geo(xr, wr, br) * ct).sum().backward(retain_graph=False)
    gx_conv = xr.grad.clone()
    want_gx = gx_conv + 3.0 * ct2
    xd, wd, bd = _dev(x), _dev(w), _dev(b)
    yd, skip = ops.dwconv_geo_skip(xd, wd, bd)
    assert skip.data_ptr() == xd.data_ptr()
    ((yd * ct.cuda()).sum() + (skip * 3.0 * ct2.cuda()).sum()).backward()
    _cmp(xd.grad, want_gx, BWD, "gx")
    _cmp(wd.grad, wr.grad, BWD, "gw")
    _cmp(bd.grad, br.grad, BWD, "gb")
    # bit-identical to the two-pass form (plain data gradient, then the addition)
    x2, w2 = _dev(x), _dev(w)
    (ops.dwconv_geo(x2, w2, _dev(b)) * ct.cuda()).sum().backward()
    assert torch.equal(xd.grad, x2.grad + 3.0 * ct2.cuda())
    # only one of the two outputs used
    x3 = _dev(x)
    y3, s3 = ops.dwconv_geo_skip(x3, _dev(w), None)
    (s3 * ct2.cuda()).sum().backward()
    assert torch.equal(x3.grad.cpu(), ct2)
    x4 = _dev(x)
    y4, _ = ops.dwconv_geo_skip(x4, _dev(w), _dev(b))
    (y4 * ct.cuda()).sum().backward()
    assert torch.equal(x4.grad, x2.grad)


@pytest.mark.parametrize("B,C,H,W,k", [(2, 6, 32, 64, 5), (5, 3, 16, 64, 5), (3, 1030, 32, 64, 5), (2, 5, 33, 64, 5),
                                       (2, 6, 12, 16, 3),
                                       # larger grids: several tiles, ragged grids (the last tile row / column overlaps
                                       # its neighbour), pole rows in different tiles than the rows they mirror onto,
                                       # many channels, one item per workgroup and many, W not a multiple of 4 (the
                                       # one-tile-per-workgroup kernels)
                                       (2, 6, 128, 256, 5), (1, 3, 70, 130, 5), (3, 5, 65, 64, 5), (2, 2100, 40, 72, 5),
                                       (1, 4, 181, 360, 5), (9, 3, 64, 128, 5), (1, 2, 721, 1440, 5), (2, 3, 32, 68, 5), (0, 4, 64, 128, 5)])
@pytest.mark.parametrize("add", [False, True])
@pytest.mark.parametrize("bias", [False, True])
def test_dwconv_geo_bwd_one_pass_is_bit_identical_to_the_two_calls(ops, B, C, H, W, k, add, bias):
    """paradis_dwconv_geo_bwd against dgrad (+ addend) and wgrad asked for apart - which tests/test_dwconv_geo pins to the
    oracle: same bits.  Whole-plane grids: ONE kernel reading the cotangent once; larger grids with k = 5 and rows of
    whole float4: the staged-tiles kernel, whose halves are what the standalone entry points run; elsewhere the
    one-tile-per-workgroup kernels."""
    gy, x, w = seeded(1, B, C, H, W).cuda(), seeded(2, B, C, H, W).cuda(), seeded(3, C, 1, k, k, scale=1.0 / k).cuda()
    ad = seeded(4, B, C, H, W).cuda() if add else None
    gx, gw, gb = ops._dwconv_geo_bwd(gy, x, w, ad, bias)
    gx2 = ops._dwconv_geo_dgrad_add(gy, w, ad) if add else ops._dwconv_geo_dgrad(gy, w)
    gw2, gb2 = ops._dwconv_geo_wgrad(gy, x, k, bias)
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)
    assert gb.numel() == (C if bias else 0)


def test_sepconv_golden(ops):
    g = load_golden("g3_blocks.pt")
    for k in (5, 7):
        rec = g[f"sepconv_k{k}"]
        P = {n: _dev(v) for n, v in rec["params"].items()}
        x = _dev(rec["x"])
        y = ops.pointwise(ops.dwconv_geo(x, P["depthwise.weight"]), P["pointwise.weight"], P["pointwise.bias"])
        y.backward(rec["cot"].cuda())
        _cmp(y, rec["y"], FWD, "y")
        _cmp(x.grad, rec["gx"], BWD, "gx")
        for n in P:
            _cmp(P[n].grad, rec["grads"][n], BWD, n)


# ----------------------------------------------------------------------------------- norm / bias
@pytest.mark.parametrize("B,C,H,W", [(2, 20, 12, 16), (2, 1024, 8, 16), (1, 1152, 9, 16), (1, 1300, 4, 8)])
def test_channel_norm(ops, B, C, H, W):
    x = seeded(1, B, C, H, W, scale=3.0) + 1.0
    w = 1.0 + seeded(2, C, scale=0.2)
    b = seeded(3, C, scale=0.2)
    ct = seeded(4, B, C, H, W)
    ts = [t.clone().requires_grad_(True) for t in (x, w, b)]
    y = O.channel_norm(*ts)
    y.backward(ct)
    ds = [_dev(t) for t in (x, w, b)]
    yd = ops.channel_norm(*ds)
    yd.backward(ct.cuda())
    _cmp(yd, y, FWD, "y")
    for n, d, t in zip(("gx", "gw", "gb"), ds, ts):
        _cmp(d.grad, t.grad, BWD, n)


def test_channel_norm_skip_output_fuses_residual_gradient(ops):
    """(y, x) = channel_norm_skip(x): the gradient arriving through the second output (the residual
    branch around a block) is added inside the backward kernel; same numbers as norm + autograd add."""
    B, C, H, W, Ce = 2, 96, 8, 16, 24
    x, xe = seeded(1, B, C, H, W), seeded(2, B, Ce, H, W)
    w, b = seeded(3, C + Ce, scale=0.5) + 1.0, seeded(4, C + Ce, scale=0.1)
    ct_y, ct_s = seeded(5, B, C + Ce, H, W), seeded(6, B, C, H, W)
    xr, xer, wr, br = (t.clone().requires_grad_(True) for t in (x, xe, w, b))
    y = O.channel_norm(torch.cat([xr, xer], 1), wr, br)
    (y * ct_y).sum().add((xr * ct_s).sum()).backward()
    xd, xed, wd, bd = (_dev(t) for t in (x, xe, w, b))
    yd, skip = ops.channel_norm_skip(xd, wd, bd, 1e-5, xed)
    assert skip.data_ptr() == xd.data_ptr()
    ((yd * ct_y.cuda()).sum() + (skip * ct_s.cuda()).sum()).backward()
    _cmp(yd, y, FWD, "y")
    for name, d, r in (("gx", xd, xr), ("gx_extra", xed, xer), ("gw", wd, wr), ("gb", bd, br)):
        _cmp(d.grad, r.grad, BWD, name)
    # only the skip output used: the gradient passes straight through
    xd2 = _dev(x)
    _, skip2 = ops.channel_norm_skip(xd2, _dev(w[:C]), _dev(b[:C]), 1e-5, None)
    (skip2 * ct_s.cuda()).sum().backward()
    assert torch.equal(xd2.grad.cpu(), ct_s)


@pytest.mark.parametrize("C1,C2,H,W", [(160, 32, 9, 16), (1024, 128, 5, 8), (160, 31, 9, 16), (161, 31, 8, 16)])
def test_channel_norm_virtual_concat_wide(ops, C1, C2, H, W):
    """the virtual concat at the widths of the reaction block: even C1, C2 take the two-workgroups-per-CU forward
    kernel (scalar row bases, a ragged last pixel tile at 9x16), odd ones the generic kernel"""
    x1, x2 = seeded(1, 2, C1, H, W, scale=2.0) + 0.5, seeded(2, 2, C2, H, W)
    C = C1 + C2
    w, b = 1.0 + seeded(3, C, scale=0.1), seeded(4, C, scale=0.1)
    ct = seeded(5, 2, C, H, W)
    r1, r2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    y = O.channel_norm(torch.cat([r1, r2], 1), w, b)
    y.backward(ct)
    d1, d2 = _dev(x1), _dev(x2)
    yd = ops.channel_norm(d1, w.cuda(), b.cuda(), 1e-5, d2)
    yd.backward(ct.cuda())
    _cmp(yd, y, FWD, "y")
    _cmp(d1.grad, r1.grad, BWD, "gx1")
    _cmp(d2.grad, r2.grad, BWD, "gx2")


def test_channel_norm_virtual_concat_and_golden(ops):
    x1, x2 = seeded(1, 2, 24, 8, 16), seeded(2, 2, 8, 8, 16, scale=2.0)
    w, b = 1.0 + seeded(3, 32, scale=0.1), seeded(4, 32, scale=0.1)
    ct = seeded(5, 2, 32, 8, 16)
    r1, r2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    y = O.channel_norm(torch.cat([r1, r2], 1), w, b)
    y.backward(ct)
    d1, d2 = _dev(x1), _dev(x2)
    yd = ops.channel_norm(d1, w.cuda(), b.cuda(), 1e-5, d2)
    yd.backward(ct.cuda())
    _cmp(yd, y, FWD, "y")
    _cmp(d1.grad, r1.grad, BWD, "gx1")
    _cmp(d2.grad, r2.grad, BWD, "gx2")
    rec = load_golden("g3_blocks.pt")["channelnorm"]
    x, wp, bp = _dev(rec["x"]), _dev(rec["params"]["weight"]), _dev(rec["params"]["bias"])
    yd = ops.channel_norm(x, wp, bp)
    yd.backward(rec["cot"].cuda())
    _cmp(yd, rec["y"], FWD, "gold y")
    _cmp(x.grad, rec["gx"], BWD, "gold gx")
    _cmp(wp.grad, rec["grads"]["weight"], BWD, "gold gw")


def test_global_bias_golden(ops):
    g = load_golden("g3_blocks.pt")
    for tag in ("noproj", "proj"):
        rec = g[f"globalbias_{tag}"]
        P = {n: _dev(v) for n, v in rec["params"].items()}
        x = _dev(rec["x"])
        bm = ops.global_bias_map(P["A"], P["U"], P["V"], P.get("projection.weight"))
        y = ops.add_bias_map(x, bm)
        y.backward(rec["cot"].cuda())
        _cmp(y, rec["y"], FWD, "y")
        _cmp(x.grad, rec["gx"], BWD, "gx")
        for n in P:
            _cmp(P[n].grad, rec["grads"][n], BWD, n)


# ----------------------------------------------------------------------------------- resampling
def test_downsample_upsample_golden(ops):
    g = load_golden("g3_blocks.pt")
    for key, rec in g.items():
        if key.startswith("downsample"):
            s = int(key.split("_s")[1])
            x = _dev(rec["x"])
            y = ops.avgpool_geo(x, s)
        elif key.startswith("upsample"):
            nlat, nlon = map(int, key.split("_")[1].split("x"))
            x = _dev(rec["x"])
            y = ops.upsample_lonp(x, nlat, nlon)
        else:
            continue
        y.backward(rec["cot"].cuda())
        _cmp(y, rec["y"], FWD, key)
        _cmp(x.grad, rec["gx"], BWD, key + " gx")
    x = torch.randn(1, 2, 8, 16)
    assert torch.equal(ops.upsample_lonp(x.cuda(), 8, 16).cpu(), x)   # identity at stride 1


@pytest.mark.parametrize("Hc,Wc,H,W", [(16, 32, 32, 64), (9, 16, 33, 64), (11, 22, 32, 66), (8, 16, 32, 64), (5, 7, 5, 28)])
def test_upsample_backward_gather_vs_oracle(ops, Hc, Wc, H, W):
    """The adjoint of the lon-periodic upsampling (reference model/paradis.py:208-220) as a gather over the fine points
    that reference a coarse cell: equal to fp64 autograd through the oracle for strides 2, 3, 4 (incl. the wrap of the
    last fine columns onto coarse column 0 and grids with pole rows), and bitwise reproducible - no atomics."""
    from oracle import paradis_oracle as O
    x = torch.randn(2, 3, Hc, Wc)
    ct = torch.randn(2, 3, H, W)
    xr = x.double().requires_grad_(True)
    O.upsample_lon_periodic(xr, H, W).backward(ct.double())
    grads = []
    for _ in range(2):
        xd = x.cuda().requires_grad_(True)
        y = ops.upsample_lonp(xd, H, W)
        y.backward(ct.cuda())
        grads.append(xd.grad.cpu())
    assert torch.equal(grads[0], grads[1])
    err = float((grads[0].double() - xr.grad).abs().max() / xr.grad.abs().max())
    assert err <= 2e-6, err
    assert float((y.detach().cpu().double() - O.upsample_lon_periodic(x.double(), H, W)).abs().max()) <= 1e-5


# ----------------------------------------------------------------------------------- elementwise
@pytest.mark.parametrize("act", ["SiLU", "GELU"])
def test_activation_and_blend(ops, act):
    x = seeded(1, 2, 5, 7, 10, scale=3.0)
    ct = seeded(2, 2, 5, 7, 10)
    xr = x.clone().requires_grad_(True)
    O.activation(xr, act).backward(ct)
    xd = _dev(x)
    yd = ops.activation(xd, act)
    yd.backward(ct.cuda())
    _cmp(yd, O.activation(x, act), FWD, "act")
    _cmp(xd.grad, xr.grad, BWD, "act grad")
    h, adv, al = seeded(3, 2, 5, 8, 16), seeded(4, 2, 5, 8, 16), seeded(5, 5)
    ct = seeded(6, 2, 5, 8, 16)
    ts = [t.clone().requires_grad_(True) for t in (h, adv, al)]
    y = ts[0] + torch.sigmoid(ts[2]).view(1, -1, 1, 1) * (ts[1] - ts[0])
    y.backward(ct)
    ds = [_dev(t) for t in (h, adv, al)]
    yd = ops.gated_blend(*ds)
    yd.backward(ct.cuda())
    _cmp(yd, y, FWD, "blend")
    for n, d, t in zip(("gh", "gadv", "galpha"), ds, ts):
        _cmp(d.grad, t.grad, BWD, n)


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 10, 7, 12, 16), (2, 130, 258, 32, 64), (1, 768, 1024, 32, 64)])
@pytest.mark.parametrize("act", [None, "SiLU"])
@pytest.mark.parametrize("scheme", ["exact", "bf16x3"])
def test_gated_epilogue_equals_gemm_then_blend(ops, B, Ci, Co, H, W, act, scheme):
    """``pointwise(..., residual=h, gate=alpha)`` = ``gated_blend(h, pointwise(...), alpha)`` (reference
    model/paradis.py:236-243) without the advected tensor: the forward bit for bit (interior and edge tiles), the
    gradients of x, W, bias and h bit for bit as well (same kernels on the same d adv), d alpha - taken against the
    blended output instead of adv - within the gradient tolerance of the oracle formula."""
    sc = {"exact": ops.GEMM_EXACT, "bf16x3": ops.GEMM_BF16X3}[scheme]
    x, w, b = seeded(1, B, Ci, H, W), seeded(2, Co, Ci, 1, 1, scale=Ci ** -0.5), seeded(3, Co, scale=0.1)
    h, al, ct = seeded(4, B, Co, H, W), seeded(5, Co), seeded(6, B, Co, H, W)
    a = [_dev(t) for t in (x, w, b, h, al)]
    y1 = ops.gated_blend(a[3], ops.pointwise(a[0], a[1], a[2], act=act, scheme=sc), a[4])
    y1.backward(ct.cuda())
    f = [_dev(t) for t in (x, w, b, h, al)]
    y2 = ops.pointwise(f[0], f[1], f[2], residual=f[3], act=act, scheme=sc, gate=f[4])
    y2.backward(ct.cuda())
    assert torch.equal(y1, y2)
    for n, p, q in zip(("gx", "gw", "gb", "gh"), a[:4], f[:4]):
        assert torch.equal(p.grad, q.grad), n
    _cmp(f[4].grad, a[4].grad.cpu(), 2e-5, "galpha")
    # against the oracle formula in fp64
    t = [v.double().requires_grad_(True) for v in (x, w, b, h, al)]
    z = torch.nn.functional.conv2d(t[0], t[1], t[2])
    z = torch.nn.functional.silu(z) if act == "SiLU" else z
    yo = t[3] + torch.sigmoid(t[4]).view(1, -1, 1, 1) * (z - t[3])
    yo.backward(ct.double())
    _cmp(y2, yo.float(), 2e-5, "y")
    _cmp(f[4].grad, t[4].grad.float(), BWD, "galpha vs fp64")
    _cmp(f[3].grad, t[3].grad.float(), BWD, "gh vs fp64")


def test_advection_transport_equals_advection_then_blend(ops):
    """NeuralSemiLagrangian.transport (blend in the up-projection's epilogue, its gradient of h through the
    down-projection's stencil) against forward_velocities + ops.gated_blend: same output bits, same gradients."""
    from paradis_model_amd.config import reduced_config
    from paradis_model_amd.harness import make_grids
    from paradis_model_amd.model.advection import NeuralSemiLagrangian
    cfg = reduced_config()
    _, lg, og = make_grids(16, 32, False)
    torch.manual_seed(3)
    adv = NeuralSemiLagrangian(cfg, 24, (16, 32), num_vels=6, lat_grid=lg, lon_grid=og,
                               interpolation=cfg.model.adv_interpolation).cuda()
    hsrc, vel, al, ct = seeded(1, 2, 24, 16, 32), seeded(2, 2, 12, 16, 32, scale=0.5), seeded(3, 24), seeded(4, 2, 24, 16, 32)
    outs = []
    for fused in (False, True):
        adv.zero_grad(set_to_none=True)
        h, v, a = _dev(hsrc), _dev(vel), _dev(al)
        h2 = h * 1.0            # a non-leaf input, as inside the model
        if fused:
            y = adv.transport(h2, v, 0.1, a)
        else:
            y = ops.gated_blend(h2, adv.forward_velocities(h2, v, 0.1), a)
        y.backward(ct.cuda())
        outs.append((y.detach(), h.grad, v.grad, a.grad, [p.grad.clone() for p in adv.parameters()]))
    (y0, gh0, gv0, ga0, gp0), (y1, gh1, gv1, ga1, gp1) = outs
    assert torch.equal(y0, y1)
    assert max_rel(gh1.cpu(), gh0.cpu()) <= 1e-6 and max_rel(gv1.cpu(), gv0.cpu()) <= 1e-6
    assert max_rel(ga1.cpu(), ga0.cpu()) <= 2e-5
    for p, q in zip(gp0, gp1):
        assert max_rel(q.cpu(), p.cpu()) <= 1e-6


def test_gmblock_golden():
    from paradis_model_amd.model import GMBlock
    rec = load_golden("g3_blocks.pt")["gmblock"]
    blk = GMBlock(layers=["CLinear", "SepConv", "CLinear"], input_dim=10, output_dim=6,
                  mesh_size=(12, 16), hidden_dim=12, bias_channels=4, pre_normalize=True,
                  activation_fn=torch.nn.GELU)
    assert list(blk.state_dict().keys()) == rec["keys"]
    blk.load_state_dict(rec["params"], strict=True)
    blk.cuda()
    x = _dev(rec["x"])
    y = blk(x)
    y.backward(rec["cot"].cuda())
    _cmp(y, rec["y"], 2e-5, "y")
    _cmp(x.grad, rec["gx"], BWD, "gx")
    for n, p in blk.named_parameters():
        _cmp(p.grad, rec["grads"][n], 2e-4, n)


@pytest.mark.parametrize("scheme", ["exact", "bf16x3", "f16x2"])
def test_empty_batch_block_ops(ops, scheme):
    """B = 0 through the block-level ops on every GEMM arithmetic: empty outputs, ZERO (not uninitialised) parameter
    gradients."""
    sc = {"exact": ops.GEMM_EXACT, "bf16x3": ops.GEMM_BF16X3, "f16x2": ops.GEMM_F16X2}[scheme]
    H, W, Ci, Co = 16, 32, 12, 20
    x = torch.zeros(0, Ci, H, W, device="cuda", requires_grad=True)
    w, b = _dev(seeded(1, Co, Ci, 1, 1)), _dev(seeded(2, Co))
    res = torch.zeros(0, Co, H, W, device="cuda", requires_grad=True)
    gate = _dev(seeded(3, Co))
    y = ops.pointwise(x, w, b, residual=res, act="SiLU", scheme=sc, gate=gate)
    assert tuple(y.shape) == (0, Co, H, W)
    y.sum().backward()
    for t in (w, b, gate):
        assert t.grad is not None and float(t.grad.abs().max()) == 0.0
    assert tuple(x.grad.shape) == (0, Ci, H, W) and tuple(res.grad.shape) == (0, Co, H, W)
    # depthwise stencil (+ skip), ChannelNorm over a virtual concat, blend, resampling
    dw = _dev(seeded(4, Ci, 1, 5, 5))
    x2 = torch.zeros(0, Ci, H, W, device="cuda", requires_grad=True)
    yd, skip = ops.dwconv_geo_skip(x2, dw, None)
    (yd.sum() + skip.sum()).backward()
    assert float(dw.grad.abs().max()) == 0.0 and tuple(x2.grad.shape) == (0, Ci, H, W)
    nw, nb = _dev(seeded(5, Ci + 4)), _dev(seeded(6, Ci + 4))
    x3 = torch.zeros(0, Ci, H, W, device="cuda", requires_grad=True)
    xe = torch.zeros(0, 4, H, W, device="cuda", requires_grad=True)
    yn = ops.channel_norm(x3, nw, nb, x_extra=xe)
    yn.sum().backward()
    assert float(nw.grad.abs().max()) == 0.0 and float(nb.grad.abs().max()) == 0.0
    al = _dev(seeded(7, Ci))
    h0 = torch.zeros(0, Ci, H, W, device="cuda", requires_grad=True)
    yb = ops.gated_blend(h0, torch.zeros(0, Ci, H, W, device="cuda", requires_grad=True), al)
    yb.sum().backward()
    assert float(al.grad.abs().max()) == 0.0
    assert tuple(ops.avgpool_geo(h0, 2).shape)[:2] == (0, Ci)
    assert tuple(ops.upsample_lonp(h0, 2 * H - 1, 2 * W).shape) == (0, Ci, 2 * H - 1, 2 * W)
