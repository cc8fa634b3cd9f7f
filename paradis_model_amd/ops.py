"""Autograd-aware Python entry points over the C-ABI HIP kernels.

Every function launches hand-written gfx950 kernels from ``libparadis_hip.so`` on
the current HIP stream (no host sync).  Inputs must be fp32 HIP tensors; there is
no CPU path (``_lib.require_hip`` raises).  Reference call sites are cited per op.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, dptr, lib, require_hip, stream_ptr

ACT_CODES = {None: 0, "none": 0, "SiLU": 1, "GELU": 2}
MODE_CODES = {"bilinear": 1, "bicubic": 2}


def _ws(nbytes: int, device) -> Optional[torch.Tensor]:
    if nbytes <= 0:
        return None
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def _planes_hw(x: torch.Tensor) -> Tuple[int, int, int]:
    H, W = x.shape[-2:]
    return x.numel() // (H * W), H, W


# ---------------------------------------------------------------------------
# a1 geocyclic padding (reference model/padding.py:11-39)
# ---------------------------------------------------------------------------
class _GeoPad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        require_hip(x)
        x = x.contiguous()
        planes, H, W = _planes_hw(x)
        y = torch.empty(*x.shape[:-2], H + 2 * p, W + 2 * p, dtype=x.dtype, device=x.device)
        check(lib.paradis_geocyclic_pad_fwd(dptr(x), dptr(y), planes, H, W, p, stream_ptr()),
              "geocyclic_pad_fwd")
        ctx.p = p
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        gx = torch.empty(ctx.shape, dtype=gy.dtype, device=gy.device)
        planes, H, W = _planes_hw(gx)
        check(lib.paradis_geocyclic_pad_bwd(dptr(gy), dptr(gx), planes, H, W, ctx.p, stream_ptr()),
              "geocyclic_pad_bwd")
        return gx, None


def geocyclic_pad(x: torch.Tensor, p: int) -> torch.Tensor:
    if p == 0:
        return x
    assert x.dim() == 4, "Input must be 4-dimensional [batch, channels, lat, lon]"
    assert x.shape[-1] % 2 == 0, "Number of longitude points must be even"
    return _GeoPad.apply(x, p)


# ---------------------------------------------------------------------------
# a3-a5 semi-Lagrangian advection core (reference model/advection.py:129-169)
# ---------------------------------------------------------------------------
class AdvectGeometry:
    """Device tables + scalars derived from the lat/lon grids once per module
    (the reference's non-persistent buffers, model/advection.py:58-72)."""

    def __init__(self, lat_grid: torch.Tensor, lon_grid: torch.Tensor):
        lat = lat_grid.detach().to(torch.float32).cpu().contiguous()
        lon = lon_grid.detach().to(torch.float32).cpu().contiguous()
        self.H, self.W = lat.shape
        # tables are evaluated once on the host in fp32, as the reference's CPU path would
        self.sin_lat = torch.sin(lat).contiguous()
        self.cos_lat = torch.cos(lat).contiguous()
        self.lon = lon
        self.min_lat = float(lat.min())
        self.min_lon = float(lon.min())
        self.d_lat = float(lat.max() - lat.min())
        self.d_lon = float(lon.max() - lon.min())
        self._dev = {}

    def tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(t.to(device) for t in (self.sin_lat, self.cos_lat, self.lon))
        return self._dev[key]


def _bstride_view(t: torch.Tensor, K: int, H: int, W: int) -> Tuple[torch.Tensor, int]:
    """Accept [B,K,H,W] views whose planes are contiguous (e.g. a channel slice); return
    (tensor, batch stride in elements)."""
    if t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W:
        return t, t.stride(0) if t.shape[0] > 1 else K * H * W
    t = t.contiguous()
    return t, K * H * W


class _SLAdvect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, field, u, v, geom: AdvectGeometry, dt: float, mode: str):
        require_hip(field, u, v)
        B, K, H, W = field.shape
        assert (H, W) == (geom.H, geom.W) and u.shape == field.shape and v.shape == field.shape
        field, f_bs = _bstride_view(field, K, H, W)
        u, u_bs = _bstride_view(u, K, H, W)
        v, v_bs = _bstride_view(v, K, H, W)
        if u_bs != v_bs:
            u, v = u.contiguous(), v.contiguous()
            u_bs = v_bs = K * H * W
        out = torch.empty(B, K, H, W, dtype=field.dtype, device=field.device)
        sl, cl, lo = geom.tables(field.device)
        ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W), field.device)
        _lib.call("sl_advect_fwd", 16.0 * B * K * H * W,   # algorithmic bytes: 16 B / gather point
                  dptr(field), dptr(u), dptr(v), dptr(out), dptr(sl), dptr(cl), dptr(lo), B, K, H, W,
                  f_bs, u_bs, K * H * W, dt, geom.min_lat, geom.min_lon, geom.d_lat, geom.d_lon,
                  MODE_CODES[mode], dptr(ws), stream_ptr())
        ctx.save_for_backward(field, u, v)
        ctx.meta = (geom, dt, mode, f_bs, u_bs)
        return out

    @staticmethod
    def backward(ctx, gout):
        field, u, v = ctx.saved_tensors
        geom, dt, mode, f_bs, u_bs = ctx.meta
        B, K, H, W = gout.shape
        gout = gout.contiguous()
        gfield = torch.empty(B, K, H, W, dtype=gout.dtype, device=gout.device)
        guv = torch.empty(B, 2 * K, H, W, dtype=gout.dtype, device=gout.device)
        gu, gv = guv[:, :K], guv[:, K:]
        sl, cl, lo = geom.tables(gout.device)
        ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W), gout.device)
        P = K * H * W
        _lib.call("sl_advect_bwd", 28.0 * B * K * H * W,   # algorithmic bytes: 28 B / gather point
                  dptr(gout), dptr(field), dptr(u), dptr(v), dptr(gfield), dptr(gu), dptr(gv), dptr(sl),
                  dptr(cl), dptr(lo), B, K, H, W, P, f_bs, u_bs, P, 2 * P, dt, geom.min_lat,
                  geom.min_lon, geom.d_lat, geom.d_lon, MODE_CODES[mode], dptr(ws), stream_ptr())
        return gfield, gu, gv, None, None, None


class _SLAdvectVel(torch.autograd.Function):
    """Same operator taking the velocity tensor [B,2K,H,W] whole (channels [0,K) = u, [K,2K) = v,
    reference model/paradis.py:236-237): no slice views, and the velocity gradient is written in
    place into one [B,2K,H,W] tensor."""

    @staticmethod
    def forward(ctx, field, vel, geom: AdvectGeometry, dt: float, mode: str):
        require_hip(field, vel)
        B, K, H, W = field.shape
        assert vel.shape == (B, 2 * K, H, W) and (H, W) == (geom.H, geom.W)
        field, f_bs = _bstride_view(field, K, H, W)
        vel = vel.contiguous()
        P = K * H * W
        out = torch.empty(B, K, H, W, dtype=field.dtype, device=field.device)
        sl, cl, lo = geom.tables(field.device)
        ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W), field.device)
        u, v = vel[:, :K], vel[:, K:]
        _lib.call("sl_advect_fwd", 16.0 * B * K * H * W, dptr(field), dptr(u), dptr(v), dptr(out), dptr(sl),
                  dptr(cl), dptr(lo), B, K, H, W, f_bs, 2 * P, P, dt, geom.min_lat, geom.min_lon, geom.d_lat,
                  geom.d_lon, MODE_CODES[mode], dptr(ws), stream_ptr())
        ctx.save_for_backward(field, vel)
        ctx.meta = (geom, dt, mode, f_bs)
        return out

    @staticmethod
    def backward(ctx, gout):
        field, vel = ctx.saved_tensors
        geom, dt, mode, f_bs = ctx.meta
        B, K, H, W = gout.shape
        gout = gout.contiguous()
        P = K * H * W
        gfield = torch.empty(B, K, H, W, dtype=gout.dtype, device=gout.device)
        gvel = torch.empty_like(vel)
        sl, cl, lo = geom.tables(gout.device)
        ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W), gout.device)
        _lib.call("sl_advect_bwd", 28.0 * B * K * H * W, dptr(gout), dptr(field), dptr(vel[:, :K]),
                  dptr(vel[:, K:]), dptr(gfield), dptr(gvel[:, :K]), dptr(gvel[:, K:]), dptr(sl), dptr(cl),
                  dptr(lo), B, K, H, W, P, f_bs, 2 * P, P, 2 * P, dt, geom.min_lat, geom.min_lon, geom.d_lat,
                  geom.d_lon, MODE_CODES[mode], dptr(ws), stream_ptr())
        return gfield, gvel, None, None, None


def sl_advect_vel(field, vel, geom: AdvectGeometry, dt: float, mode: str = "bicubic"):
    if mode not in MODE_CODES:
        raise ValueError(f"interpolation must be one of {list(MODE_CODES)}")
    return _SLAdvectVel.apply(field, vel, geom, float(dt), mode)


def sl_advect(field, u, v, geom: AdvectGeometry, dt: float, mode: str = "bicubic"):
    """[B,K,H,W] x3 -> [B,K,H,W]; fused pole-mean / departure / gather / pole-mean."""
    if mode not in MODE_CODES:
        raise ValueError(f"interpolation must be one of {list(MODE_CODES)}")
    return _SLAdvect.apply(field, u, v, geom, float(dt), mode)


# ---------------------------------------------------------------------------
# helpers for [B,C,H,W] tensors whose (H,W) planes are contiguous
# ---------------------------------------------------------------------------
def _plane_view(t: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """Return (tensor, batch stride in elements); channel slices of a larger contiguous tensor are
    accepted as they are, anything else is made contiguous."""
    B, C, H, W = t.shape
    if t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W:
        return t, (t.stride(0) if B > 1 else C * H * W)
    return t.contiguous(), C * H * W


# ---------------------------------------------------------------------------
# a7 depthwise stencil on the virtual geocyclic halo (reference model/blocks.py:101-113)
# ---------------------------------------------------------------------------
class _DwConvGeo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        require_hip(x, weight, bias)
        x = x.contiguous()
        B, C, H, W = x.shape
        k = weight.shape[-1]
        assert weight.shape == (C, 1, k, k), "depthwise weight must be [C,1,k,k]"
        w = weight.contiguous()
        y = torch.empty_like(x)
        check(lib.paradis_dwconv_geo_fwd(dptr(x), dptr(w), dptr(bias), dptr(y), B, C, H, W, k,
                                         stream_ptr()), "dwconv_geo_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, C, H, W = x.shape
        k = w.shape[-1]
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            check(lib.paradis_dwconv_geo_dgrad(dptr(gy), dptr(w), dptr(gx), B, C, H, W, k, stream_ptr()),
                  "dwconv_geo_dgrad")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(w)
            gb = torch.empty(C, dtype=x.dtype, device=x.device) if ctx.has_bias else None
            ws = _ws(lib.paradis_dwconv_geo_wgrad_ws_bytes(B, C, H, W, k), x.device)
            check(lib.paradis_dwconv_geo_wgrad(dptr(gy), dptr(x), dptr(gw), dptr(gb), B, C, H, W, k,
                                               dptr(ws), stream_ptr()), "dwconv_geo_wgrad")
        return gx, gw, gb


def dwconv_geo(x, weight, bias=None):
    return _DwConvGeo.apply(x, weight, bias)


# ---------------------------------------------------------------------------
# a11 / a14 resampling
# ---------------------------------------------------------------------------
class _AvgPoolGeo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride):
        require_hip(x)
        x = x.contiguous()
        planes, H, W = _planes_hw(x)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty(*x.shape[:-2], Ho, Wo, dtype=x.dtype, device=x.device)
        check(lib.paradis_avgpool_geo_fwd(dptr(x), dptr(y), planes, H, W, stride, stream_ptr()),
              "avgpool_geo_fwd")
        ctx.meta = (x.shape, stride)
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, stride = ctx.meta
        gy = gy.contiguous()
        gx = torch.empty(shape, dtype=gy.dtype, device=gy.device)
        planes, H, W = _planes_hw(gx)
        check(lib.paradis_avgpool_geo_bwd(dptr(gy), dptr(gx), planes, H, W, stride, stream_ptr()),
              "avgpool_geo_bwd")
        return gx, None


_BOX_WEIGHTS = {}


def avgpool_geo(x, stride: int):
    if stride < 1:
        raise ValueError("Coarsening factor must be >=1")
    if stride == 1 and x.is_cuda:
        # stride 1 = depthwise 5x5 stencil with uniform taps 1/25: reuse the LDS-tiled kernels
        # (forward 2.8x, backward 5x faster than the generic strided kernels)
        key = (x.shape[1], x.device)
        if key not in _BOX_WEIGHTS:
            _BOX_WEIGHTS[key] = torch.full((x.shape[1], 1, 5, 5), 1.0 / 25.0, dtype=torch.float32,
                                           device=x.device)
        return _DwConvGeo.apply(x, _BOX_WEIGHTS[key], None)
    return _AvgPoolGeo.apply(x, int(stride))


class _UpsampleLonP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nlat, nlon):
        require_hip(x)
        x = x.contiguous()
        planes, Hc, Wc = _planes_hw(x)
        y = torch.empty(*x.shape[:-2], nlat, nlon, dtype=x.dtype, device=x.device)
        check(lib.paradis_upsample_lonp_fwd(dptr(x), dptr(y), planes, Hc, Wc, nlat, nlon, stream_ptr()),
              "upsample_lonp_fwd")
        ctx.meta = (x.shape, nlat, nlon)
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, nlat, nlon = ctx.meta
        gy = gy.contiguous()
        gx = torch.empty(shape, dtype=gy.dtype, device=gy.device)
        planes, Hc, Wc = _planes_hw(gx)
        check(lib.paradis_upsample_lonp_bwd(dptr(gy), dptr(gx), planes, Hc, Wc, nlat, nlon, stream_ptr()),
              "upsample_lonp_bwd")
        return gx, None, None


def upsample_lonp(x, nlat: int, nlon: int):
    if tuple(x.shape[-2:]) == (int(nlat), int(nlon)):
        return x   # equal sizes: ATen's align_corners interpolation is the exact identity
    return _UpsampleLonP.apply(x, int(nlat), int(nlon))


# ---------------------------------------------------------------------------
# a8 ChannelNorm (reference model/blocks.py:118-134), optionally over a virtual concat
# ---------------------------------------------------------------------------
class _ChannelNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, weight, bias, eps, with_skip):
        require_hip(x1, x2, weight, bias)
        x1_in = x1
        x1, bs1 = _plane_view(x1)
        B, C1, H, W = x1.shape
        C2, bs2 = 0, 0
        if x2 is not None:
            x2, bs2 = _plane_view(x2)
            C2 = x2.shape[1]
        P = H * W
        C = C1 + C2
        y = torch.empty(B, C, H, W, dtype=x1.dtype, device=x1.device)
        mean = torch.empty(B, P, dtype=x1.dtype, device=x1.device)
        rstd = torch.empty_like(mean)
        check(lib.paradis_channel_norm_fwd(dptr(x1), dptr(x2), dptr(weight), dptr(bias), dptr(y),
                                           dptr(mean), dptr(rstd), B, C1, C2, P, bs1, bs2, eps,
                                           stream_ptr()), "channel_norm_fwd")
        ctx.save_for_backward(x1, x2 if x2 is not None else x1.new_empty(0), weight, mean, rstd)
        ctx.meta = (C1, C2, bs1, bs2, H, W)
        if with_skip:
            # second output = x1 itself: consumers of the residual path take it from here, so both
            # gradients of x1 arrive at this node and are summed inside the backward kernel
            ctx.set_materialize_grads(False)
            return y, x1_in
        return y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x1, x2, weight, mean, rstd = ctx.saved_tensors
        C1, C2, bs1, bs2, H, W = ctx.meta
        B, P, C = x1.shape[0], H * W, C1 + C2
        if gy is None:          # only the skip path was used
            return gskip, None, None, None, None, None
        gy = gy.contiguous()
        add, add_bs = None, 0
        if gskip is not None:
            add, add_bs = _plane_view(gskip)
        gx1 = torch.empty(B, C1, H, W, dtype=gy.dtype, device=gy.device)
        gx2 = torch.empty(B, C2, H, W, dtype=gy.dtype, device=gy.device) if C2 else None
        gw = torch.empty(C, dtype=gy.dtype, device=gy.device)
        gb = torch.empty_like(gw)
        ws = _ws(lib.paradis_channel_norm_bwd_ws_bytes(B, C, P), gy.device)
        check(lib.paradis_channel_norm_bwd(dptr(gy), dptr(x1), dptr(x2) if C2 else None, dptr(weight),
                                           dptr(mean), dptr(rstd), dptr(gx1), dptr(gx2), dptr(gw),
                                           dptr(gb), B, C1, C2, P, bs1, bs2, C1 * P, C2 * P, dptr(add),
                                           add_bs, dptr(ws), stream_ptr()), "channel_norm_bwd")
        return gx1, gx2, gw, gb, None, None


def channel_norm(x, weight, bias, eps: float = 1e-5, x_extra=None):
    """ChannelNorm over channels of ``x`` (and, virtually concatenated after them, ``x_extra``)."""
    return _ChannelNorm.apply(x, x_extra, weight, bias, float(eps), False)


def channel_norm_skip(x, weight, bias, eps: float = 1e-5, x_extra=None):
    """``(channel_norm(x), x)``: the second output is ``x`` for the residual branch around the block.
    Its gradient is added to the normalisation's input gradient inside the backward kernel instead of
    by a separate autograd accumulation pass."""
    return _ChannelNorm.apply(x, x_extra, weight, bias, float(eps), True)


# ---------------------------------------------------------------------------
# a9 GlobalBias map (reference model/blocks.py:188-196)
# ---------------------------------------------------------------------------
class _GlobalBiasMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, U, V, Pw):
        require_hip(A, U, V, Pw)
        A, U, V = A.contiguous(), U.contiguous(), V.contiguous()
        Cin, R = A.shape
        H, W = U.shape[1], V.shape[1]
        if Pw is not None:
            Pw = Pw.contiguous()
            Co = Pw.shape[0]
            m8 = torch.empty(Cin, H, W, dtype=A.dtype, device=A.device)
        else:
            Co, m8 = Cin, None
        out = torch.empty(Co, H, W, dtype=A.dtype, device=A.device)
        check(lib.paradis_global_bias_map_fwd(dptr(A), dptr(U), dptr(V), dptr(Pw), dptr(m8), dptr(out),
                                              Cin, Co, R, H, W, stream_ptr()), "global_bias_map_fwd")
        ctx.save_for_backward(A, U, V, Pw if Pw is not None else A.new_empty(0),
                              m8 if m8 is not None else A.new_empty(0))
        ctx.has_proj = Pw is not None
        return out

    @staticmethod
    def backward(ctx, gmap):
        A, U, V, Pw, m8 = ctx.saved_tensors
        Cin, R = A.shape
        H, W = U.shape[1], V.shape[1]
        gmap = gmap.contiguous()
        Co = gmap.shape[0]
        gA, gU, gV = torch.empty_like(A), torch.empty_like(U), torch.empty_like(V)
        gPw = torch.empty_like(Pw) if ctx.has_proj else None
        ws = _ws(lib.paradis_global_bias_map_bwd_ws_bytes(Cin, Co, R, H, W), A.device)
        check(lib.paradis_global_bias_map_bwd(dptr(gmap), dptr(A), dptr(U), dptr(V),
                                              dptr(Pw) if ctx.has_proj else None,
                                              dptr(m8) if ctx.has_proj else None, dptr(gA), dptr(gU),
                                              dptr(gV), dptr(gPw), Cin, Co, R, H, W, dptr(ws),
                                              stream_ptr()), "global_bias_map_bwd")
        return gA, gU, gV, gPw


def global_bias_map(A, U, V, Pw=None):
    return _GlobalBiasMap.apply(A, U, V, Pw)


class _GlobalBiasM8(torch.autograd.Function):
    """Un-projected rank-R map m8[Cin,H,W]; the projection to the layer width is applied inside
    the GEMM epilogue (``pointwise(..., bias_proj=(m8, Pw))``) so the [Co,H,W] map is never stored."""

    @staticmethod
    def forward(ctx, A, U, V):
        require_hip(A, U, V)
        A, U, V = A.contiguous(), U.contiguous(), V.contiguous()
        Cin, R = A.shape
        H, W = U.shape[1], V.shape[1]
        m8 = torch.empty(Cin, H, W, dtype=A.dtype, device=A.device)
        check(lib.paradis_global_bias_m8_fwd(dptr(A), dptr(U), dptr(V), dptr(m8), Cin, R, H, W, stream_ptr()),
              "global_bias_m8_fwd")
        ctx.save_for_backward(A, U, V)
        return m8

    @staticmethod
    def backward(ctx, gm8):
        A, U, V = ctx.saved_tensors
        Cin, R = A.shape
        H, W = U.shape[1], V.shape[1]
        gm8 = gm8.contiguous()
        gA, gU, gV = torch.empty_like(A), torch.empty_like(U), torch.empty_like(V)
        ws = _ws(lib.paradis_global_bias_map_bwd_ws_bytes(Cin, Cin, R, H, W), A.device)
        check(lib.paradis_global_bias_m8_bwd(dptr(gm8), dptr(A), dptr(U), dptr(V), dptr(gA), dptr(gU), dptr(gV),
                                             Cin, R, H, W, dptr(ws), stream_ptr()), "global_bias_m8_bwd")
        return gA, gU, gV


def global_bias_m8(A, U, V):
    return _GlobalBiasM8.apply(A, U, V)


# ---------------------------------------------------------------------------
# a6 pointwise channel mixing on FP32 MFMA with fused epilogue
#     y = residual + act(W x + bias + bias_map)
# (reference model/blocks.py:86,110 + :196 + activation + the residual adds of paradis.py:246,253)
# ---------------------------------------------------------------------------
# Arithmetic of the pointwise GEMMs (include/paradis_hip.h, a6): True = bf16-split products on the
# bf16 matrix pipe (fp32 in/accumulate/out, error vs fp64 not above the exact path's: see
# tests/test_hip_gemm_split.py), False = exact f32 MFMA chain.  PARADIS_GEMM=exact selects the latter.
GEMM_SPLIT = os.environ.get("PARADIS_GEMM", "split") != "exact"


def _split_weights(w2: torch.Tensor, Co: int, Ci: int, transpose: bool) -> torch.Tensor:
    nbytes = lib.paradis_pw_gemm_split_bytes(Ci, Co) if transpose else lib.paradis_pw_gemm_split_bytes(Co, Ci)
    out = torch.empty(nbytes, dtype=torch.uint8, device=w2.device)
    check(lib.paradis_pw_gemm_split_weights(dptr(w2), Co, Ci, 1 if transpose else 0, dptr(out), stream_ptr()),
          "pw_gemm_split_weights")
    return out


class _Pointwise(torch.autograd.Function):
    """y = residual + act(W x + bias + bias_map).

    Activation-gradient hand-off between two chained ops (GMBlock drives it):
      * ``defer_act_grad`` (producer): the op's backward receives d(pre-activation) directly and
        skips its own act' pass; it also returns its pre-activation ``z`` as a second output.
      * ``x_pre`` / ``x_act`` (consumer): x = act(x_pre) was produced by such an op; the consumer's
        dgrad multiplies by act'(x_pre) in the GEMM epilogue, so what it returns as the gradient of
        ``x`` already is the producer's d(pre-activation).
    """

    @staticmethod
    def forward(ctx, x, weight, bias, bmap, residual, act, x_pre, x_act, defer_act_grad, m8, pw):
        require_hip(x, weight, bias, bmap, residual, x_pre, m8, pw)
        x, x_bs = _plane_view(x)
        B, Ci, H, W = x.shape
        Co = weight.shape[0]
        P = H * W
        w2 = weight.reshape(Co, -1)
        assert w2.shape[1] == Ci, "weight/in-channel mismatch"
        w2 = w2.contiguous()
        res_bs = 0
        if residual is not None:
            residual, res_bs = _plane_view(residual)
        if bmap is not None:
            bmap = bmap.contiguous()
        y = torch.empty(B, Co, H, W, dtype=x.dtype, device=x.device)
        need_z = act != 0 and (any(ctx.needs_input_grad[:4]) or any(ctx.needs_input_grad[9:11])
                               or defer_act_grad)
        cin = 0
        if pw is not None:
            m8, pw = m8.contiguous(), pw.contiguous()
            cin = pw.shape[1]
            assert pw.shape[0] == Co and m8.shape == (cin, H, W) and bmap is None
            assert Co % 4 == 0 and cin <= 16, "fused GlobalBias projection needs Co % 4 == 0 and <= 16 bias channels"
            pwt = pw.t().contiguous()   # [cin, Co]: four consecutive output rows per 16-byte load
        z = torch.empty_like(y) if need_z else None
        w2t = wsp = None
        split = GEMM_SPLIT
        if split:
            # bf16-split image of the weights (h/m/l planes in tile order) for the split-MFMA kernel
            wsp = _split_weights(w2, Co, Ci, transpose=False)
        elif Ci % 16 == 0 and Co % 4 == 0 and Co * Ci >= 4096:
            # [Ci,Co] copy of the weights: makes the A operand row-contiguous for the LDS-DMA kernel
            w2t = torch.empty(Ci, Co, dtype=x.dtype, device=x.device)
            check(lib.paradis_transpose(dptr(w2), dptr(w2t), Co, Ci, stream_ptr()), "transpose")
        _lib.call("pw_gemm_fwd", 2.0 * B * Co * Ci * P, dptr(w2), dptr(w2t), dptr(wsp), dptr(x), dptr(bias),
                  dptr(bmap), dptr(m8), dptr(pwt) if cin else None, cin, dptr(residual), dptr(y), dptr(z), B, Co, Ci, P, x_bs,
                  res_bs, Co * P, act, stream_ptr())
        if x_pre is not None:
            x_pre = x_pre.contiguous()
        ctx.save_for_backward(x, w2, z if z is not None else x.new_empty(0),
                              x_pre if x_pre is not None else x.new_empty(0),
                              m8 if pw is not None else x.new_empty(0),
                              pw if pw is not None else x.new_empty(0))
        ctx.meta = (x_bs, act, bias is not None, bmap is not None, residual is not None, weight.shape,
                    x_act if x_pre is not None else 0, bool(defer_act_grad), split)
        if defer_act_grad:
            # z carries no gradient; without this autograd would materialise a full-size zero tensor
            # for it on every backward (24 fills of [B,896,H,W] per training step)
            ctx.mark_non_differentiable(z)
            ctx.set_materialize_grads(False)
            return y, z
        return y

    @staticmethod
    def backward(ctx, gy, *unused):
        x, w2, z, x_pre, m8, pw = ctx.saved_tensors
        x_bs, act, has_bias, has_map, has_res, wshape, x_act, deferred, split = ctx.meta
        has_proj = pw.numel() > 0
        B, Ci, H, W = x.shape
        Co, P = w2.shape[0], H * W
        gy = gy.contiguous()
        st = stream_ptr()
        gres = gy if has_res else None
        if act != 0 and not deferred:
            dz = torch.empty_like(gy)
            check(lib.paradis_act_bwd(dptr(gy), dptr(z), dptr(dz), gy.numel(), act, st), "act_bwd")
        else:
            dz = gy          # no activation, or the consumer already applied act'(z) (deferred)
        gx = gw = gb = gmap = gm8 = gpw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(B, Ci, H, W, dtype=gy.dtype, device=gy.device)
            zmul = x_pre if x_act != 0 else None
            wtsp = _split_weights(w2, Co, Ci, transpose=True) if split else None
            _lib.call("pw_gemm_dgrad", 2.0 * B * Co * Ci * P, dptr(w2), dptr(wtsp), dptr(dz), dptr(zmul), None, dptr(gx),
                      B, Co, Ci, P, Co * P, Ci * P, 0, Ci * P, x_act, st)
        want_b = has_bias and ctx.needs_input_grad[2]
        want_p = has_proj and (ctx.needs_input_grad[9] or ctx.needs_input_grad[10])
        want_m = (has_map and ctx.needs_input_grad[3]) or want_p
        if ctx.needs_input_grad[1]:
            gw = torch.empty(Co, Ci, dtype=gy.dtype, device=gy.device)
            ws = _ws(lib.paradis_pw_gemm_wgrad_ws_bytes(B, Co, Ci, P), gy.device)
            if want_b and not want_m:   # bias gradient = row sums of dz: fused into the wgrad GEMM
                gb = torch.empty(Co, dtype=gy.dtype, device=gy.device)
                want_b = False
            _lib.call("pw_gemm_wgrad", 2.0 * B * Co * Ci * P, dptr(dz), dptr(x), dptr(gw), dptr(gb), B, Co,
                      Ci, P, Co * P, x_bs, 1 if split else 0, dptr(ws), st)
            gw = gw.reshape(wshape)
        if want_b or want_m:
            gb = torch.empty(Co, dtype=gy.dtype, device=gy.device) if want_b else None
            gmap = torch.empty(Co, H, W, dtype=gy.dtype, device=gy.device) if want_m else None
            check(lib.paradis_bias_grads(dptr(dz), dptr(gmap), dptr(gb), B, Co, P, Co * P, st), "bias_grads")
        if want_p:   # adjoint of the fused projection: gmap -> (gPw, gm8); the full map only lives here
            cin = pw.shape[1]
            gpw = torch.empty_like(pw)
            gm8 = torch.empty_like(m8)
            check(lib.paradis_global_bias_proj_bwd(dptr(gmap), dptr(m8), dptr(pw), dptr(gpw), dptr(gm8), cin,
                                                   Co, P, st), "global_bias_proj_bwd")
            gmap = None
        return gx, gw, gb, gmap, gres, None, None, None, None, gm8, gpw


def pointwise(x, weight, bias=None, bias_map=None, residual=None, act=None, x_pre=None, x_act=None,
              defer_act_grad=False, bias_proj=None):
    """y = residual + act(weight . x + bias[:,None] + bias_map); weight [Co,Ci] or [Co,Ci,1,1].

    ``bias_proj=(m8[Cin,H,W], Pw[Co,Cin])`` adds the projected low-rank GlobalBias map inside the GEMM
    epilogue instead of a materialised ``bias_map``.
    ``defer_act_grad=True`` returns ``(y, z)`` and expects the consumer to be another ``pointwise``
    called with ``x_pre=z, x_act=act`` (see ``_Pointwise``); only valid when ``y`` has no other use."""
    if defer_act_grad and (act is None or residual is not None):
        raise ValueError("defer_act_grad needs an activation and no residual")
    m8, pw = bias_proj if bias_proj is not None else (None, None)
    return _Pointwise.apply(x, weight, bias, bias_map, residual, ACT_CODES[act], x_pre,
                            ACT_CODES[x_act], bool(defer_act_grad), m8, pw)


# ---------------------------------------------------------------------------
# elementwise glue
# ---------------------------------------------------------------------------
class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        require_hip(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.paradis_act_fwd(dptr(x), dptr(y), x.numel(), act, stream_ptr()), "act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        check(lib.paradis_act_bwd(dptr(gy), dptr(x), dptr(gx), x.numel(), ctx.act, stream_ptr()), "act_bwd")
        return gx, None


def activation(x, name: str):
    if name not in ("SiLU", "GELU"):
        raise ValueError(f"Unknown activation_fn '{name}'. Allowed: ['SiLU', 'GELU']")
    return _Activation.apply(x, ACT_CODES[name])


class _GatedBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, adv, alpha):
        require_hip(h, adv, alpha)
        h, adv, alpha = h.contiguous(), adv.contiguous(), alpha.contiguous()
        B, C, H, W = h.shape
        out = torch.empty_like(h)
        check(lib.paradis_gated_blend_fwd(dptr(h), dptr(adv), dptr(alpha), dptr(out), B, C, H * W,
                                          stream_ptr()), "gated_blend_fwd")
        ctx.save_for_backward(h, adv, alpha)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, adv, alpha = ctx.saved_tensors
        B, C, H, W = h.shape
        gout = gout.contiguous()
        gh, gadv, galpha = torch.empty_like(h), torch.empty_like(h), torch.empty_like(alpha)
        ws = _ws(lib.paradis_gated_blend_bwd_ws_bytes(B, C, H * W), h.device)
        check(lib.paradis_gated_blend_bwd(dptr(gout), dptr(h), dptr(adv), dptr(alpha), dptr(gh),
                                          dptr(gadv), dptr(galpha), B, C, H * W, dptr(ws), stream_ptr()),
              "gated_blend_bwd")
        return gh, gadv, galpha


def gated_blend(h, adv, alpha):
    """h + sigmoid(alpha)[None,:,None,None] * (adv - h)   (reference model/paradis.py:239-243)"""
    return _GatedBlend.apply(h, adv, alpha)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        require_hip(a, b)
        assert a.shape == b.shape
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        check(lib.paradis_add(dptr(a), dptr(b), dptr(y), a.numel(), stream_ptr()), "add")
        return y

    @staticmethod
    def backward(ctx, gy):
        return gy, gy


def add(a, b):
    return _Add.apply(a, b)


class _AddBiasMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bmap):
        require_hip(x, bmap)
        x, bmap = x.contiguous(), bmap.contiguous()
        B = x.shape[0]
        assert x.shape[1:] == bmap.shape
        y = torch.empty_like(x)
        check(lib.paradis_add_bcast(dptr(x), dptr(bmap), dptr(y), bmap.numel(), B, stream_ptr()), "add_bcast")
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        B, C, H, W = gy.shape
        gmap = None
        if ctx.needs_input_grad[1]:
            gmap = torch.empty(C, H, W, dtype=gy.dtype, device=gy.device)
            check(lib.paradis_bias_grads(dptr(gy), dptr(gmap), None, B, C, H * W, C * H * W, stream_ptr()),
                  "bias_grads")
        return gy, gmap


def add_bias_map(x, bmap):
    """x[B,C,H,W] + bmap[C,H,W] (standalone GlobalBias, reference model/blocks.py:196)"""
    return _AddBiasMap.apply(x, bmap)


# ---------------------------------------------------------------------------
# rows f1-f3: loss, rollout glue, optimiser step
# ---------------------------------------------------------------------------
class _ParadisLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, wf, wl, kind, delta):
        require_hip(pred, target, wf, wl)
        pred, target = pred.contiguous(), target.contiguous()
        B, C, H, W = pred.shape
        assert target.shape == pred.shape and wf.numel() == C and (wl is None or wl.numel() == H)
        loss = torch.empty((), dtype=pred.dtype, device=pred.device)
        want_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(pred) if want_grad else None
        partial = torch.empty(lib.paradis_loss_blocks(pred.numel()), dtype=pred.dtype, device=pred.device)
        check(lib.paradis_loss_fwd_bwd(dptr(pred), dptr(target), dptr(wf), dptr(wl), dptr(loss), dptr(grad),
                                       dptr(partial), B, C, H, W, kind, delta, stream_ptr()), "loss_fwd_bwd")
        if want_grad:
            ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad)
        check(lib.paradis_scale(dptr(grad), dptr(gout.contiguous()), dptr(out), grad.numel(), stream_ptr()),
              "scale")
        return out, None, None, None, None, None


def paradis_loss(pred, target, feature_weights, lat_weights=None, kind="reversed_huber", delta=1.0):
    """mean(feature_weights[c] * lat_weights[h] * l(pred - target)); forward and d/dpred in one pass."""
    code = {"mse": 0, "reversed_huber": 1}[kind]
    return _ParadisLoss.apply(pred, target, feature_weights, lat_weights, code, float(delta))


class _ConcatChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *parts):
        require_hip(*parts)
        B = parts[0].shape[0]
        H, W = parts[0].shape[-2:]
        chans = [p.shape[1] for p in parts]
        P = H * W
        out = torch.empty(B, sum(chans), H, W, dtype=parts[0].dtype, device=parts[0].device)
        off = 0
        for p, c in zip(parts, chans):
            p, bs = _plane_view(p)
            check(lib.paradis_copy_channels(dptr(p), bs, dptr(out[:, off:]), sum(chans) * P, B, c * P,
                                            stream_ptr()), "copy_channels")
            off += c
        ctx.chans = chans
        return out

    @staticmethod
    def backward(ctx, gout):
        gout = gout.contiguous()
        B, Ct, H, W = gout.shape
        P = H * W
        grads, off = [], 0
        for i, c in enumerate(ctx.chans):
            if ctx.needs_input_grad[i]:
                g = torch.empty(B, c, H, W, dtype=gout.dtype, device=gout.device)
                check(lib.paradis_copy_channels(dptr(gout[:, off:]), Ct * P, dptr(g), c * P, B, c * P,
                                                stream_ptr()), "copy_channels")
                grads.append(g)
            else:
                grads.append(None)
            off += c
        return tuple(grads)


def concat_channels(parts):
    """cat(parts, dim=1) for [B,C_i,H,W] tensors (channel slices accepted) by strided block copies."""
    return _ConcatChannels.apply(*parts)
