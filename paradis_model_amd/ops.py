"""torch custom ops (namespace ``paradis``) over the C-ABI HIP kernels.

Every entry point of the hot path is registered through ``torch.library``:

* a ``CUDA`` (= HIP on ROCm) kernel that launches the hand-written gfx950 kernels of
  ``libparadis_hip.so`` on the current HIP stream through the C ABI (no host sync);
* a fake (Meta) kernel giving the output shapes, so ``torch.compile(fullgraph=True)`` and
  AOTAutograd trace the model with these ops as opaque nodes (reference ``trainer.py:261-267``,
  ``model/paradis.py:195-206``);
* an autograd formula whose backward is again made of ``paradis::*_backward`` ops, so the traced
  backward graph contains no Python-only calls;
* an autocast rule: the kernels are fp32, under ``torch.autocast`` (the reference's
  ``precision="bf16-mixed"``, ``train.py:56``) inputs are cast to fp32 - never a narrower
  arithmetic than the reference's.

There is no CPU kernel: a CPU tensor raises (``_lib.require_hip`` / the dispatcher), and so does any
dtype but fp32.  The Python functions below (``pointwise``, ``sl_advect`` ...) are thin wrappers that
check arguments and call ``torch.ops.paradis.*``.  Reference call sites are cited per op.
"""
from __future__ import annotations

import os
import threading
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, dptr, lib, stream_ptr

ACT_CODES = {None: 0, "none": 0, "SiLU": 1, "GELU": 2}
MODE_CODES = {"bilinear": 1, "bicubic": 2}

_DEF = torch.library.Library("paradis", "DEF")
OPS = {}            # op name -> OpOverload (torch.ops.paradis.<name>.default)
RAW = {}            # op name -> the Python function behind its HIP kernel (eager backward formulas call it directly)


# (private torch entry points, resolved once: if a torch build lacks them nothing is ever "plain" and every call takes
#  the dispatcher - slower, never wrong)
_dispatch_modes = getattr(torch._C, "_len_torch_dispatch_stack", None)
_functorch_top = getattr(getattr(torch._C, "_functorch", None), "peek_interpreter_stack", None)
_CAN_BE_PLAIN = _dispatch_modes is not None and _functorch_top is not None


def _plain(t) -> bool:
    """a real device tensor in an eager, un-traced call: the kernels' Python functions may be called directly (fake /
    functional / subclass tensors - FakeTensorMode, make_fx, torch.export - must go through the dispatcher)"""
    return (_CAN_BE_PLAIN and type(t) is Tensor and not torch.is_grad_enabled()
            and _dispatch_modes() == 0                    # make_fx(tracing_mode="real"), any TorchDispatchMode
            and _functorch_top() is None)                 # vmap / functional grad: plain-looking wrappers


def _all_plain(args) -> bool:
    """every tensor argument is a real ``Tensor`` / ``Parameter`` (no subclass or wrapper whose ``__torch_dispatch__`` the
    direct call would skip) and nothing watches the dispatcher"""
    if not (_CAN_BE_PLAIN and _dispatch_modes() == 0 and _functorch_top() is None):
        return False
    for a in args:
        if isinstance(a, Tensor) and type(a) is not Tensor and type(a) is not torch.nn.Parameter:
            return False
    return True


def _widen_for_autocast(args):
    """Inside ``torch.autocast`` the ops' autocast rule widens bf16 / fp16 tensor arguments to fp32 (the kernels are fp32).
    An eager ``autograd.Function`` front end sits ABOVE that rule, so it applies the rule itself before the kernel call
    AND before it saves anything for backward - a narrow saved input would reach the fp32-only backward kernels
    (tests/test_hip_amp.py::test_narrow_inputs_under_autocast_in_grad_mode).  Autograd casts the returned
    gradients back to the inputs' dtypes."""
    if not torch.is_autocast_enabled("cuda"):
        return args
    return tuple(a.float() if (isinstance(a, Tensor) and a.is_floating_point() and a.dtype != torch.float32) else a
                 for a in args)


def _eager_forward(name: str, op, *args):
    """inside an ``autograd.Function.forward`` of an eager front end: the kernel's Python function directly where nothing
    watches the dispatcher and every tensor argument is a plain fp32 tensor, else the registered op"""
    if not torch.is_grad_enabled() and _all_plain(args) and (name == "pointwise" or not torch.is_autocast_enabled("cuda")):
        return RAW[name](*args)
    return op(*args)


def _selective_autocast(name: str, keep16: Tuple[int, ...], scheme_idx: Optional[int]) -> None:
    """Autocast rule of the pointwise-GEMM ops (round 6): widen every floating tensor argument to fp32 EXCEPT the
    bf16-stored activations / gradients at the positions ``keep16`` when the call's scheme (argument ``scheme_idx``; none
    = always) is the bf16-mixed one - there a bf16 tensor is the storage format the kernels consume, not something
    autocast has to undo (torch.library.register_autocast can only cast everything)."""
    op = getattr(torch.ops.paradis, name).default
    keyset = (torch._C.DispatchKeySet(torch._C.DispatchKey.AutocastCPU)
              | torch._C.DispatchKeySet(torch._C.DispatchKey.AutocastCUDA))

    def cast(args):
        mixed = scheme_idx is None or args[scheme_idx] == GEMM_BF16
        return tuple(a.float() if (isinstance(a, Tensor) and a.is_floating_point() and a.dtype != torch.float32
                                   and not (mixed and i in keep16 and a.dtype == torch.bfloat16)) else a
                     for i, a in enumerate(args))

    def py_impl(*args, **kwargs):
        assert not kwargs, "custom ops take positional arguments here"
        with torch._C._ExcludeDispatchKeyGuard(keyset):
            return op(*cast(args))

    try:
        op.py_impl(torch._C.DispatchKey.AutocastCUDA)(py_impl)
    except Exception:                      # (a torch without py_impl on OpOverload: the C++-side kernel below suffices)
        pass
    _DEF.impl(name, lambda _ks, *args, **kwargs: py_impl(*args, **kwargs), "AutocastCUDA", with_keyset=True)


def _define(schema: str, autocast=True):
    """Register ``schema`` in the ``paradis`` namespace with the decorated function as its HIP kernel.
    ``autocast``: True = every floating tensor argument is widened to fp32 under torch.autocast; a tuple
    ``(keep16 positions, scheme position or None)`` = the selective rule of ``_selective_autocast``; False = none."""
    name = schema[: schema.index("(")]
    _DEF.define(schema)

    def deco(fn):
        RAW[name] = fn
        _DEF.impl(name, fn, "CUDA")
        if autocast is True:
            torch.library.register_autocast(f"paradis::{name}", "cuda", torch.float32)
        elif autocast:
            _selective_autocast(name, tuple(autocast[0]), autocast[1])
        ov = getattr(torch.ops.paradis, name).default
        OPS[name] = ov
        return ov
    return deco


def _fake(name: str):
    """Register the fake (shape) kernel of an op."""
    return torch.library.register_fake(f"paradis::{name}")


def _autograd(name: str, setup, backward) -> None:
    torch.library.register_autograd(f"paradis::{name}", backward, setup_context=setup)


def require_hip(*tensors) -> None:
    """Device check of the Python wrappers (the dtype check sits in the kernels, behind the autocast
    rule that widens bf16/fp16 inputs to fp32)."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("paradis_model_amd ops run on the MI355X only: got a CPU tensor "
                               "(there is no CPU fallback; use oracle/ for CPU checks)")


def _f32(*tensors) -> None:
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"paradis_model_amd ops are fp32; got {t.dtype}")


def _f32_or_bf16(scheme: int, *tensors) -> None:
    """tensors that may be STORED as bf16 in the bf16-mixed scheme (round 6): GEMM activations, their pre-activations
    and the gradients of both; fp32 everywhere else"""
    for t in tensors:
        if t is not None and t.dtype != torch.float32 and not (t.dtype == torch.bfloat16 and scheme == GEMM_BF16):
            raise RuntimeError(f"paradis_model_amd ops are fp32 (bf16 tensors only in the bf16-mixed GEMM scheme); got {t.dtype}")


def _is16(t) -> bool:
    return t is not None and t.dtype == torch.bfloat16


# ---- bf16-stored activations between a NON-GEMM producer and its pointwise consumer (round 6) ----------------------------
# ChannelNorm and the depthwise stencil can write their output as a bf16 tensor when its only consumer is a pointwise GEMM
# in the bf16-mixed scheme (``out_bf16``).  Autograd hands a bf16 tensor's producer a bf16 cotangent - the consumer's data
# gradient, bf16-valued in the reference's autocast backward as well (the gradient of conv2d's bf16 input).  The producers'
# backward kernels read it as it is stored where they have a bf16 instantiation (the streaming ChannelNorm backward, the
# whole-plane stencil backward: ``paradis_*_bwd16``); on every other path the op widens it first.  No side channel: the
# cotangent is an ordinary tensor, so checkpointing, hooks and foreign producers / consumers need no special case.
def _cotangent(gy, takes_bf16: bool):
    """the cotangent of a (possibly bf16-stored) output as the backward kernels of this node take it"""
    if gy is None or gy.dtype == torch.float32 or (takes_bf16 and gy.dtype == torch.bfloat16):
        return gy
    return gy.float()


def _want_bf16_out(x: Tensor, out_bf16: bool) -> bool:
    """a producer's ``out_bf16`` request is honoured where its consumer - a pointwise GEMM called next, inside the same
    autocast region - will run the bf16-mixed scheme and take a bf16 operand (see ``pointwise``)"""
    return (bool(out_bf16) and BF16_STORAGE and autocast_scheme(GEMM_SCHEME) == GEMM_BF16
            and (x.shape[-2] * x.shape[-1]) % 16 == 0 and not torch.compiler.is_compiling())


def _ws(nbytes: int, device) -> Optional[Tensor]:
    if nbytes <= 0:
        return None
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def _planes_hw(x: Tensor) -> Tuple[int, int, int]:
    H, W = x.shape[-2:]
    return x.numel() // (H * W), H, W


def _none_if_empty(t: Tensor) -> Optional[Tensor]:
    return t if t.numel() else None


def _plane_view(t: Tensor) -> Tuple[Tensor, int]:
    """[B,C,H,W] whose (H,W) planes are contiguous and channels are P apart (e.g. a channel slice of a
    larger contiguous tensor) is consumed in place: return (tensor, batch stride in elements);
    anything else is made contiguous."""
    B, C, H, W = t.shape
    if t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W:
        return t, (t.stride(0) if B > 1 else C * H * W)
    return t.contiguous(), C * H * W


# ---------------------------------------------------------------------------
# a1 geocyclic padding (reference model/padding.py:11-39)
# ---------------------------------------------------------------------------
@_define("geocyclic_pad(Tensor x, int p) -> Tensor")
def _geocyclic_pad(x, p):
    _f32(x)
    x = x.contiguous()
    planes, H, W = _planes_hw(x)
    y = torch.empty(*x.shape[:-2], H + 2 * p, W + 2 * p, dtype=x.dtype, device=x.device)
    check(lib.paradis_geocyclic_pad_fwd(dptr(x), dptr(y), planes, H, W, p, stream_ptr()), "geocyclic_pad_fwd")
    return y


@_fake("geocyclic_pad")
def _(x, p):
    return x.new_empty(*x.shape[:-2], x.shape[-2] + 2 * p, x.shape[-1] + 2 * p)


@_define("geocyclic_pad_backward(Tensor gy, int p) -> Tensor")
def _geocyclic_pad_backward(gy, p):
    _f32(gy)
    gy = gy.contiguous()
    H, W = gy.shape[-2] - 2 * p, gy.shape[-1] - 2 * p
    gx = torch.empty(*gy.shape[:-2], H, W, dtype=gy.dtype, device=gy.device)
    check(lib.paradis_geocyclic_pad_bwd(dptr(gy), dptr(gx), gx.numel() // (H * W), H, W, p, stream_ptr()),
          "geocyclic_pad_bwd")
    return gx


@_fake("geocyclic_pad_backward")
def _(gy, p):
    return gy.new_empty(*gy.shape[:-2], gy.shape[-2] - 2 * p, gy.shape[-1] - 2 * p)


def _pad_setup(ctx, inputs, output):
    ctx.p = inputs[1]


def _pad_backward(ctx, gy):
    return _geocyclic_pad_backward(gy, ctx.p), None


_autograd("geocyclic_pad", _pad_setup, _pad_backward)


def geocyclic_pad(x: Tensor, p: int) -> Tensor:
    if p == 0:
        return x
    assert x.dim() == 4, "Input must be 4-dimensional [batch, channels, lat, lon]"
    assert x.shape[-1] % 2 == 0, "Number of longitude points must be even"
    require_hip(x)
    return _geocyclic_pad(x, int(p))


# ---------------------------------------------------------------------------
# a3-a5 semi-Lagrangian advection core (reference model/advection.py:129-169)
# ---------------------------------------------------------------------------
class AdvectGeometry:
    """Device tables + scalars derived from the lat/lon grids once per module
    (the reference's non-persistent buffers, model/advection.py:58-72)."""

    def __init__(self, lat_grid: Tensor, lon_grid: Tensor):
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
        # regular lat-lon grid: latitude depends on the row only, longitude on the column only
        self.separable = bool((lat == lat[:, :1]).all()) and bool((lon == lon[:1, :]).all())
        # arrival latitude in cells, (lat - min_lat) (H-1)/d_lat, in double; the padding p is added per mode
        cells = (lat.double() - self.min_lat) * ((self.H - 1.0) / self.d_lat) if self.d_lat > 0 else lat.double() * 0
        self.lat_cells = {p: (cells + float(p)).to(torch.float32).contiguous() for p in (1, 2)}
        self._dev = {}

    def tables(self, device, p: int = 2):
        """(sin_lat, cos_lat, lat_cells, lon) on ``device``; lat_cells = p + (lat - min_lat)(H-1)/d_lat as fp32"""
        key = (str(device), p)
        if key not in self._dev:
            self._dev[key] = tuple(t.to(device) for t in (self.sin_lat, self.cos_lat, self.lat_cells[p], self.lon))
        return self._dev[key]


_ADV_TABLES = "Tensor sin_lat, Tensor cos_lat, Tensor lat_cells, Tensor lon"
_ADV_GEOM = "float dt, float min_lat, float min_lon, float d_lat, float d_lon, int mode, int flags"


def _bstride_view(t: Tensor, K: int, H: int, W: int) -> Tuple[Tensor, int]:
    if t.stride(3) == 1 and t.stride(2) == W and t.stride(1) == H * W:
        return t, t.stride(0) if t.shape[0] > 1 else K * H * W
    return t.contiguous(), K * H * W


@_define(f"sl_advect(Tensor field, Tensor u, Tensor v, {_ADV_TABLES}, {_ADV_GEOM}) -> Tensor")
def _sl_advect(field, u, v, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    _f32(field, u, v)
    B, K, H, W = field.shape
    field, f_bs = _bstride_view(field, K, H, W)
    u, u_bs = _bstride_view(u, K, H, W)
    v, v_bs = _bstride_view(v, K, H, W)
    if u_bs != v_bs:
        u, v = u.contiguous(), v.contiguous()
        u_bs = K * H * W
    out = torch.empty(B, K, H, W, dtype=field.dtype, device=field.device)
    ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W, flags), field.device)
    _lib.call("sl_advect_fwd", 16.0 * B * K * H * W,   # algorithmic bytes: 16 B / gather point
              dptr(field), dptr(u), dptr(v), dptr(out), dptr(sl), dptr(cl), dptr(lc), dptr(lo), B, K, H, W,
              f_bs, u_bs, K * H * W, dt, min_lat, min_lon, d_lat, d_lon, mode, flags, dptr(ws), stream_ptr())
    return out


@_fake("sl_advect")
def _(field, u, v, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    return field.new_empty(field.shape)


@_define(f"sl_advect_backward(Tensor gout, Tensor field, Tensor u, Tensor v, {_ADV_TABLES}, "
         f"{_ADV_GEOM}) -> (Tensor, Tensor, Tensor)")
def _sl_advect_backward(gout, field, u, v, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    _f32(gout, field, u, v)
    B, K, H, W = gout.shape
    P = K * H * W
    gout = gout.contiguous()
    field, f_bs = _bstride_view(field, K, H, W)
    u, u_bs = _bstride_view(u, K, H, W)
    v, v_bs = _bstride_view(v, K, H, W)
    if u_bs != v_bs:
        u, v = u.contiguous(), v.contiguous()
        u_bs = P
    gfield = torch.empty(B, K, H, W, dtype=gout.dtype, device=gout.device)
    gu, gv = torch.empty_like(gfield), torch.empty_like(gfield)
    ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W, flags), gout.device)
    _lib.call("sl_advect_bwd", 28.0 * B * K * H * W,   # algorithmic bytes: 28 B / gather point
              dptr(gout), dptr(field), dptr(u), dptr(v), dptr(gfield), dptr(gu), dptr(gv), dptr(sl),
              dptr(cl), dptr(lc), dptr(lo), B, K, H, W, P, f_bs, u_bs, P, P, dt, min_lat, min_lon, d_lat, d_lon, mode,
              flags, dptr(ws), stream_ptr())
    return gfield, gu, gv


@_fake("sl_advect_backward")
def _(gout, field, u, v, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    return gout.new_empty(gout.shape), gout.new_empty(gout.shape), gout.new_empty(gout.shape)


def _adv_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:7])
    ctx.geom = inputs[7:]


def _adv_backward(ctx, gout):
    gf, gu, gv = _sl_advect_backward(gout, *ctx.saved_tensors, *ctx.geom)
    return (gf, gu, gv) + (None,) * 11


_autograd("sl_advect", _adv_setup, _adv_backward)


@_define(f"sl_advect_vel(Tensor field, Tensor vel, {_ADV_TABLES}, {_ADV_GEOM}) -> Tensor")
def _sl_advect_vel(field, vel, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    """Same operator taking the velocity tensor [B,2K,H,W] whole (channels [0,K) = u, [K,2K) = v,
    reference model/paradis.py:236-237): no slice views, and the velocity gradient is written in
    place into one [B,2K,H,W] tensor."""
    _f32(field, vel)
    B, K, H, W = field.shape
    field, f_bs = _bstride_view(field, K, H, W)
    vel = vel.contiguous()
    P = K * H * W
    out = torch.empty(B, K, H, W, dtype=field.dtype, device=field.device)
    ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W, flags), field.device)
    _lib.call("sl_advect_fwd", 16.0 * B * K * H * W, dptr(field), dptr(vel[:, :K]), dptr(vel[:, K:]), dptr(out),
              dptr(sl), dptr(cl), dptr(lc), dptr(lo), B, K, H, W, f_bs, 2 * P, P, dt, min_lat, min_lon, d_lat, d_lon, mode,
              flags, dptr(ws), stream_ptr())
    return out


@_fake("sl_advect_vel")
def _(field, vel, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    return field.new_empty(field.shape)


@_define(f"sl_advect_vel_backward(Tensor gout, Tensor field, Tensor vel, {_ADV_TABLES}, "
         f"{_ADV_GEOM}) -> (Tensor, Tensor)")
def _sl_advect_vel_backward(gout, field, vel, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    _f32(gout, field, vel)
    B, K, H, W = gout.shape
    P = K * H * W
    gout = gout.contiguous()
    field, f_bs = _bstride_view(field, K, H, W)
    vel = vel.contiguous()
    gfield = torch.empty(B, K, H, W, dtype=gout.dtype, device=gout.device)
    gvel = torch.empty_like(vel)
    ws = _ws(lib.paradis_sl_advect_ws_bytes(B, K, H, W, flags), gout.device)
    _lib.call("sl_advect_bwd", 28.0 * B * K * H * W, dptr(gout), dptr(field), dptr(vel[:, :K]),
              dptr(vel[:, K:]), dptr(gfield), dptr(gvel[:, :K]), dptr(gvel[:, K:]), dptr(sl), dptr(cl),
              dptr(lc), dptr(lo), B, K, H, W, P, f_bs, 2 * P, P, 2 * P, dt, min_lat, min_lon, d_lat, d_lon, mode,
              flags, dptr(ws), stream_ptr())
    return gfield, gvel


@_fake("sl_advect_vel_backward")
def _(gout, field, vel, sl, cl, lc, lo, dt, min_lat, min_lon, d_lat, d_lon, mode, flags):
    return gout.new_empty(gout.shape), vel.new_empty(vel.shape)


def _advv_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:6])
    ctx.geom = inputs[6:]


def _advv_backward(ctx, gout, raw=False):
    saved = ctx.saved_tensors        # once: non-reentrant checkpointing unpacks (recomputes) on this access
    if raw is None:
        raw = _plain(gout) and _all_plain(saved)
    gf, gvel = (RAW if raw else OPS)["sl_advect_vel_backward"](gout, *saved, *ctx.geom)
    return (gf, gvel) + (None,) * 11


_autograd("sl_advect_vel", _advv_setup, _advv_backward)


ADVECT_GENERIC, ADVECT_TILED, ADVECT_SEPARABLE, ADVECT_TILES, ADVECT_STRIPS = 1, 2, 4, 8, 16     # include/paradis_hip.h PARADIS_ADVECT_*
# schedule hints applied when a call passes none (diagnostics: tools/advect_halo_sweep.py)
ADVECT_FLAGS = int(os.environ.get("PARADIS_ADVECT_FLAGS", "0"), 0)


def advect_flags(tiled: bool = False, halo: Optional[int] = None, halo_bwd: Optional[int] = None,
                 generic: bool = False, tiles: bool = False, strips: bool = False) -> int:
    """``flags`` of the advection ops: force the windowed schedule and/or its longitude halo (padded cells;
    ``halo_bwd`` overrides it for the backward kernel), the generic whole-plane kernel, or (``tiles``, diagnostic)
    the tile schedule of rounds 2-3 in place of the strips; ``strips``: the backward's 128-column strips where the
    full-circle ring (W <= 256) would run."""
    return ((ADVECT_TILED if tiled else 0) | (ADVECT_GENERIC if generic else 0) | (ADVECT_TILES if tiles else 0)
            | (ADVECT_STRIPS if strips else 0)
            | (((int(halo) + 1) << 8) if halo is not None else 0)
            | (((int(halo_bwd) + 1) << 16) if halo_bwd is not None else 0))


def _geom_args(geom: AdvectGeometry, device, dt: float, mode: str, flags: Optional[int]):
    if mode not in MODE_CODES:
        raise ValueError(f"interpolation must be one of {list(MODE_CODES)}")
    sl, cl, lc, lo = geom.tables(device, 2 if mode == "bicubic" else 1)
    flags = ADVECT_FLAGS if flags is None else int(flags)
    if geom.separable:
        flags |= ADVECT_SEPARABLE
    return sl, cl, lc, lo, float(dt), geom.min_lat, geom.min_lon, geom.d_lat, geom.d_lon, MODE_CODES[mode], flags


class _AdvectVelEager(torch.autograd.Function):
    """``paradis::sl_advect_vel`` with its registered setup / backward, for eager recording calls (see ``_PointwiseEager``)."""

    @staticmethod
    def forward(ctx, *args):
        args = _widen_for_autocast(args)
        out = _eager_forward("sl_advect_vel", _sl_advect_vel, *args)
        _advv_setup(ctx, args, out)
        return out

    @staticmethod
    def backward(ctx, gout):
        return _advv_backward(ctx, gout, raw=None)      # None: decided on the unpacked tensors (ONE ctx.saved_tensors access)


def sl_advect_vel(field, vel, geom: AdvectGeometry, dt: float, mode: str = "bicubic", flags: Optional[int] = None):
    require_hip(field, vel)
    B, K, H, W = field.shape
    assert vel.shape == (B, 2 * K, H, W) and (H, W) == (geom.H, geom.W)
    args = (field, vel, *_geom_args(geom, field.device, dt, mode, flags))
    if torch.is_grad_enabled() and not torch.compiler.is_compiling():
        return _AdvectVelEager.apply(*args)
    return _sl_advect_vel(*args)


def sl_advect(field, u, v, geom: AdvectGeometry, dt: float, mode: str = "bicubic", flags: Optional[int] = None):
    """[B,K,H,W] x3 -> [B,K,H,W]; fused pole-mean / departure / gather / pole-mean."""
    require_hip(field, u, v)
    assert tuple(field.shape[-2:]) == (geom.H, geom.W) and u.shape == field.shape and v.shape == field.shape
    return _sl_advect(field, u, v, *_geom_args(geom, field.device, dt, mode, flags))


# ---------------------------------------------------------------------------
# a7 depthwise stencil on the virtual geocyclic halo (reference model/blocks.py:101-113)
# ---------------------------------------------------------------------------
@_define("dwconv_geo(Tensor x, Tensor weight, Tensor? bias, bool out_bf16=False) -> Tensor")
def _dwconv_geo(x, weight, bias, out_bf16=False):
    """``out_bf16``: y as a bf16 tensor (bf16-mixed mode: the consumer is the SepConv's pointwise GEMM, which rounds its
    operand to bf16 anyway - reference model/blocks.py:107-110 under autocast)."""
    _f32(x, weight, bias)
    x = x.contiguous()
    B, C, H, W = x.shape
    k = weight.shape[-1]
    assert weight.shape == (C, 1, k, k), "depthwise weight must be [C,1,k,k]"
    w = weight.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    fn = lib.paradis_dwconv_geo_fwd16 if out_bf16 else lib.paradis_dwconv_geo_fwd
    check(fn(dptr(x), dptr(w), dptr(bias), dptr(y), B, C, H, W, k, stream_ptr()), "dwconv_geo_fwd")
    return y


@_fake("dwconv_geo")
def _(x, weight, bias, out_bf16=False):
    return x.new_empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32)


@_define("dwconv_geo_dgrad(Tensor gy, Tensor weight) -> Tensor")
def _dwconv_geo_dgrad(gy, weight):
    _f32(gy, weight)
    gy, w = gy.contiguous(), weight.contiguous()
    B, C, H, W = gy.shape
    gx = torch.empty_like(gy)
    check(lib.paradis_dwconv_geo_dgrad(dptr(gy), dptr(w), dptr(gx), B, C, H, W, w.shape[-1], stream_ptr()),
          "dwconv_geo_dgrad")
    return gx


@_fake("dwconv_geo_dgrad")
def _(gy, weight):
    return gy.new_empty(gy.shape)


@_define("dwconv_geo_dgrad_add(Tensor gy, Tensor weight, Tensor addend) -> Tensor")
def _dwconv_geo_dgrad_add(gy, weight, addend):
    _f32(gy, weight, addend)
    gy, w, addend = gy.contiguous(), weight.contiguous(), addend.contiguous()
    assert addend.shape == gy.shape
    B, C, H, W = gy.shape
    gx = torch.empty_like(gy)
    check(lib.paradis_dwconv_geo_dgrad_add(dptr(gy), dptr(w), dptr(addend), dptr(gx), B, C, H, W, w.shape[-1],
                                           stream_ptr()), "dwconv_geo_dgrad_add")
    return gx


@_fake("dwconv_geo_dgrad_add")
def _(gy, weight, addend):
    return gy.new_empty(gy.shape)


@_define("dwconv_geo_wgrad(Tensor gy, Tensor x, int k, bool has_bias) -> (Tensor, Tensor)")
def _dwconv_geo_wgrad(gy, x, k, has_bias):
    _f32(gy, x)
    gy, x = gy.contiguous(), x.contiguous()
    B, C, H, W = x.shape
    gw = torch.empty(C, 1, k, k, dtype=x.dtype, device=x.device)
    gb = torch.empty(C if has_bias else 0, dtype=x.dtype, device=x.device)
    ws = _ws(lib.paradis_dwconv_geo_wgrad_ws_bytes(B, C, H, W, k), x.device)
    check(lib.paradis_dwconv_geo_wgrad(dptr(gy), dptr(x), dptr(gw), dptr(gb) if has_bias else None, B, C, H, W, k,
                                       dptr(ws), stream_ptr()), "dwconv_geo_wgrad")
    return gw, gb


@_fake("dwconv_geo_wgrad")
def _(gy, x, k, has_bias):
    C = x.shape[1]
    return x.new_empty(C, 1, k, k), x.new_empty(C if has_bias else 0)


@_define("dwconv_geo_bwd(Tensor gy, Tensor x, Tensor weight, Tensor? addend, bool has_bias) -> (Tensor, Tensor, Tensor)")
def _dwconv_geo_bwd(gy, x, weight, addend, has_bias):
    """(gx (+ addend), gw, gb) of the stencil in one call: with k = 5 on the reference grids one kernel that reads gy
    once (paradis_dwconv_geo_bwd); bit-identical to ``dwconv_geo_dgrad`` (``_add``) + ``dwconv_geo_wgrad``."""
    _f32(x, weight, addend)
    _f32_or_bf16(gy)
    B, C, H, W = x.shape
    k = weight.shape[-1]
    if _is16(gy) and not lib.paradis_dwconv_geo_bwd16_ok(H, W, k):
        gy = gy.float()             # (a bf16 cotangent is read as stored on the whole-plane grids only)
    gy, x, w = gy.contiguous(), x.contiguous(), weight.contiguous()
    if addend is not None:
        addend = addend.contiguous()
        assert addend.shape == gy.shape
    gx = torch.empty(gy.shape, dtype=x.dtype, device=x.device)
    gw = torch.empty(C, 1, k, k, dtype=x.dtype, device=x.device)
    gb = torch.empty(C if has_bias else 0, dtype=x.dtype, device=x.device)
    ws = _ws(lib.paradis_dwconv_geo_wgrad_ws_bytes(B, C, H, W, k), x.device)
    fn = lib.paradis_dwconv_geo_bwd16 if _is16(gy) else lib.paradis_dwconv_geo_bwd
    check(fn(dptr(gy), dptr(x), dptr(w), dptr(addend), dptr(gx), dptr(gw),
             dptr(gb) if has_bias else None, B, C, H, W, k, dptr(ws), stream_ptr()), "dwconv_geo_bwd")
    return gx, gw, gb


@_fake("dwconv_geo_bwd")
def _(gy, x, weight, addend, has_bias):
    C, k = x.shape[1], weight.shape[-1]
    return x.new_empty(gy.shape), x.new_empty(C, 1, k, k), x.new_empty(C if has_bias else 0)


def _dw_grads(ctx, gy, x, w, addend, raw=False):
    """input / weight / bias gradients as the context needs them (one fused call when it needs both kinds);
    ``raw``: an eager backward - the kernels' Python functions directly (see ``_pw_backward``)"""
    K = RAW if raw else OPS
    need_x = ctx.needs_input_grad[0]
    need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
    gx = gw = gb = None
    gy = _cotangent(gy, need_x and need_w)       # (the fused kernel has the bf16-cotangent instantiation)
    if need_x and need_w:
        gx, gw, gb = K["dwconv_geo_bwd"](gy, x, w, addend, ctx.has_bias)
    elif need_x:
        gx = K["dwconv_geo_dgrad"](gy, w) if addend is None else K["dwconv_geo_dgrad_add"](gy, w, addend)
    elif need_w:
        gw, gb = K["dwconv_geo_wgrad"](gy, x, w.shape[-1], ctx.has_bias)
    if not ctx.has_bias:
        gb = None
    return gx, gw, gb


def _dw_setup(ctx, inputs, output):
    x, w, bias = inputs[:3]
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _dw_backward(ctx, gy):
    x, w = ctx.saved_tensors
    return (*_dw_grads(ctx, gy, x, w, None), None)


_autograd("dwconv_geo", _dw_setup, _dw_backward)


def dwconv_geo(x, weight, bias=None, out_bf16: bool = False):
    """``out_bf16``: a request (see ``_want_bf16_out``) - the stencil's output as a bf16 tensor for a pointwise consumer"""
    require_hip(x, weight, bias)
    return _dwconv_geo(x, weight, bias, _want_bf16_out(x, out_bf16))


class _DwconvSkip(torch.autograd.Function):
    """``(dwconv_geo(x), x)``: the second output is ``x`` itself for a consumer around the block (the gated blend next
    to the advection's down-projection), so both gradients of ``x`` arrive at this node and the data-gradient kernel
    adds the other one as it stores (``paradis_dwconv_geo_dgrad_add``) - no accumulation pass of the autograd engine
    over the tensor.  Eager only, like ``_ChannelNormSkip``."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_bf16):
        y = _eager_forward("dwconv_geo", _dwconv_geo, x, weight, bias, out_bf16)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        return y, x

    @staticmethod
    def backward(ctx, gy, gskip):
        x, w = ctx.saved_tensors
        if gy is None:          # only the skip path was used
            return gskip, None, None, None
        return (*_dw_grads(ctx, gy, x, w, gskip, raw=_plain(gy)), None)


def dwconv_geo_skip(x, weight, bias=None, out_bf16: bool = False):
    """``(dwconv_geo(x), x)`` for a stencil whose input has another consumer (see ``_DwconvSkip``)."""
    require_hip(x, weight, bias)
    if torch.compiler.is_compiling():
        return _dwconv_geo(x, weight, bias), x
    return _DwconvSkip.apply(x, weight, bias, _want_bf16_out(x, out_bf16))


# ---------------------------------------------------------------------------
# a11 / a14 resampling
# ---------------------------------------------------------------------------
@_define("avgpool_geo(Tensor x, int stride) -> Tensor")
def _avgpool_geo(x, stride):
    _f32(x)
    x = x.contiguous()
    planes, H, W = _planes_hw(x)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(*x.shape[:-2], Ho, Wo, dtype=x.dtype, device=x.device)
    check(lib.paradis_avgpool_geo_fwd(dptr(x), dptr(y), planes, H, W, stride, stream_ptr()), "avgpool_geo_fwd")
    return y


@_fake("avgpool_geo")
def _(x, stride):
    H, W = x.shape[-2:]
    return x.new_empty(*x.shape[:-2], (H - 1) // stride + 1, (W - 1) // stride + 1)


@_define("avgpool_geo_backward(Tensor gy, int H, int W, int stride) -> Tensor")
def _avgpool_geo_backward(gy, H, W, stride):
    _f32(gy)
    gy = gy.contiguous()
    gx = torch.empty(*gy.shape[:-2], H, W, dtype=gy.dtype, device=gy.device)
    check(lib.paradis_avgpool_geo_bwd(dptr(gy), dptr(gx), gx.numel() // (H * W), H, W, stride, stream_ptr()),
          "avgpool_geo_bwd")
    return gx


@_fake("avgpool_geo_backward")
def _(gy, H, W, stride):
    return gy.new_empty(*gy.shape[:-2], H, W)


def _pool_setup(ctx, inputs, output):
    ctx.meta = (inputs[0].shape[-2], inputs[0].shape[-1], inputs[1])


def _pool_backward(ctx, gy):
    return _avgpool_geo_backward(gy, *ctx.meta), None


_autograd("avgpool_geo", _pool_setup, _pool_backward)

_BOX_WEIGHTS = {}


def avgpool_geo(x, stride: int):
    if stride < 1:
        raise ValueError("Coarsening factor must be >=1")
    require_hip(x)
    if stride == 1:
        # stride 1 = depthwise 5x5 stencil with uniform taps 1/25: reuse the LDS-tiled kernels
        # (forward 2.8x, backward 5x faster than the generic strided kernels)
        key = (x.shape[1], str(x.device))
        if key not in _BOX_WEIGHTS:
            _BOX_WEIGHTS[key] = torch.full((x.shape[1], 1, 5, 5), 1.0 / 25.0, dtype=torch.float32,
                                           device=x.device)
        return _dwconv_geo(x, _BOX_WEIGHTS[key], None)
    return _avgpool_geo(x, int(stride))


@_define("upsample_lonp(Tensor x, int nlat, int nlon) -> Tensor")
def _upsample_lonp(x, nlat, nlon):
    _f32(x)
    x = x.contiguous()
    planes, Hc, Wc = _planes_hw(x)
    y = torch.empty(*x.shape[:-2], nlat, nlon, dtype=x.dtype, device=x.device)
    check(lib.paradis_upsample_lonp_fwd(dptr(x), dptr(y), planes, Hc, Wc, nlat, nlon, stream_ptr()),
          "upsample_lonp_fwd")
    return y


@_fake("upsample_lonp")
def _(x, nlat, nlon):
    return x.new_empty(*x.shape[:-2], nlat, nlon)


@_define("upsample_lonp_backward(Tensor gy, int Hc, int Wc) -> Tensor")
def _upsample_lonp_backward(gy, Hc, Wc):
    _f32(gy)
    gy = gy.contiguous()
    nlat, nlon = gy.shape[-2:]
    gx = torch.empty(*gy.shape[:-2], Hc, Wc, dtype=gy.dtype, device=gy.device)
    check(lib.paradis_upsample_lonp_bwd(dptr(gy), dptr(gx), gx.numel() // (Hc * Wc), Hc, Wc, nlat, nlon,
                                        stream_ptr()), "upsample_lonp_bwd")
    return gx


@_fake("upsample_lonp_backward")
def _(gy, Hc, Wc):
    return gy.new_empty(*gy.shape[:-2], Hc, Wc)


def _up_setup(ctx, inputs, output):
    ctx.meta = tuple(inputs[0].shape[-2:])


def _up_backward(ctx, gy):
    return _upsample_lonp_backward(gy, *ctx.meta), None, None


_autograd("upsample_lonp", _up_setup, _up_backward)


def upsample_lonp(x, nlat: int, nlon: int):
    if tuple(x.shape[-2:]) == (int(nlat), int(nlon)):
        return x   # equal sizes: ATen's align_corners interpolation is the exact identity
    require_hip(x)
    return _upsample_lonp(x, int(nlat), int(nlon))


# ---------------------------------------------------------------------------
# a8 ChannelNorm (reference model/blocks.py:118-134), optionally over a virtual concat
# ---------------------------------------------------------------------------
@_define("channel_norm(Tensor x1, Tensor? x2, Tensor weight, Tensor bias, float eps, bool out_bf16=False) "
         "-> (Tensor, Tensor, Tensor)")
def _channel_norm(x1, x2, weight, bias, eps, out_bf16=False):
    """``out_bf16``: y as a bf16 tensor (bf16-mixed mode, consumer = a pointwise GEMM: the reference casts the norm's fp32
    output to bf16 at that conv2d, model/blocks.py:86 under train.py:56); mean / rstd stay fp32."""
    _f32(x1, x2, weight, bias)
    x1, bs1 = _plane_view(x1)
    B, C1, H, W = x1.shape
    C2, bs2 = 0, 0
    if x2 is not None:
        x2, bs2 = _plane_view(x2)
        C2 = x2.shape[1]
    P = H * W
    y = torch.empty(B, C1 + C2, H, W, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x1.device)
    mean = torch.empty(B, P, dtype=x1.dtype, device=x1.device)
    rstd = torch.empty_like(mean)
    fn = lib.paradis_channel_norm_fwd16 if out_bf16 else lib.paradis_channel_norm_fwd
    check(fn(dptr(x1), dptr(x2), dptr(weight), dptr(bias), dptr(y), dptr(mean),
             dptr(rstd), B, C1, C2, P, bs1, bs2, eps, stream_ptr()), "channel_norm_fwd")
    return y, mean, rstd


@_fake("channel_norm")
def _(x1, x2, weight, bias, eps, out_bf16=False):
    B, C1, H, W = x1.shape
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    return (x1.new_empty(B, C, H, W, dtype=torch.bfloat16 if out_bf16 else torch.float32), x1.new_empty(B, H * W),
            x1.new_empty(B, H * W))


@_define("channel_norm_backward(Tensor gy, Tensor x1, Tensor? x2, Tensor weight, Tensor mean, Tensor rstd, "
         "Tensor? add) -> (Tensor, Tensor, Tensor, Tensor)")
def _channel_norm_backward(gy, x1, x2, weight, mean, rstd, add):
    """``add``: a gradient that reaches x1 along another path (the residual branch around the block);
    it is added inside the kernel instead of by a separate accumulation pass."""
    _f32(x1, x2, add)
    _f32_or_bf16(gy)
    x1, bs1 = _plane_view(x1)
    B, C1, H, W = x1.shape
    C2, bs2 = 0, 0
    if x2 is not None:
        x2, bs2 = _plane_view(x2)
        C2 = x2.shape[1]
    P, C = H * W, C1 + C2
    gy = gy.contiguous()
    add_bs = 0
    if add is not None:
        add, add_bs = _plane_view(add)
    if _is16(gy) and not lib.paradis_channel_norm_bwd16_ok(B, C, P):
        gy = gy.float().contiguous()        # (a bf16 cotangent is read as stored by the streaming kernels only)
    gx1 = torch.empty(B, C1, H, W, dtype=x1.dtype, device=gy.device)
    gx2 = torch.empty(B, C2, H, W, dtype=x1.dtype, device=gy.device)
    gw = torch.empty(C, dtype=x1.dtype, device=gy.device)
    gb = torch.empty_like(gw)
    ws = _ws(lib.paradis_channel_norm_bwd_ws_bytes(B, C, P), gy.device)
    fn = lib.paradis_channel_norm_bwd16 if _is16(gy) else lib.paradis_channel_norm_bwd
    check(fn(dptr(gy), dptr(x1), dptr(x2) if C2 else None, dptr(weight), dptr(mean),
             dptr(rstd), dptr(gx1), dptr(gx2) if C2 else None, dptr(gw), dptr(gb), B,
             C1, C2, P, bs1, bs2, C1 * P, C2 * P, dptr(add), add_bs, dptr(ws),
             stream_ptr()), "channel_norm_bwd")
    return gx1, gx2, gw, gb


@_fake("channel_norm_backward")
def _(gy, x1, x2, weight, mean, rstd, add):
    B, C1, H, W = x1.shape
    C2 = x2.shape[1] if x2 is not None else 0
    return (x1.new_empty(B, C1, H, W), x1.new_empty(B, C2, H, W), x1.new_empty(C1 + C2), x1.new_empty(C1 + C2))


def _norm_setup(ctx, inputs, output):
    x1, x2, weight, bias, eps = inputs[:5]
    y, mean, rstd = output
    ctx.save_for_backward(x1, x2, weight, mean, rstd)
    ctx.mark_non_differentiable(mean, rstd)
    ctx.set_materialize_grads(False)


def _norm_backward(ctx, gy, gmean=None, grstd=None):
    x1, x2, weight, mean, rstd = ctx.saved_tensors
    if gy is None:
        return None, None, None, None, None, None
    gx1, gx2, gw, gb = _channel_norm_backward(_cotangent(gy, True), x1, x2, weight, mean, rstd, None)
    return gx1, (gx2 if x2 is not None else None), gw, gb, None, None


_autograd("channel_norm", _norm_setup, _norm_backward)


class _ChannelNormSkip(torch.autograd.Function):
    """``(channel_norm(x), x)``: the second output is ``x`` itself for the residual branch around the
    block, so every gradient of ``x`` arrives at this node and is added inside the backward kernel
    (``channel_norm_backward(..., add=...)``) instead of by autograd's accumulation pass.  Eager only:
    under ``torch.compile`` the plain op is used and the traced graph holds the addition."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, eps, out_bf16):
        y, mean, rstd = _eager_forward("channel_norm", _channel_norm, x1, x2, weight, bias, eps, out_bf16)
        ctx.save_for_backward(x1, x2, weight, mean, rstd)
        ctx.set_materialize_grads(False)
        return y, x1

    @staticmethod
    def backward(ctx, gy, gskip):
        x1, x2, weight, mean, rstd = ctx.saved_tensors
        if gy is None:          # only the skip path was used
            return gskip, None, None, None, None, None
        gy = _cotangent(gy, True)
        bwd = RAW["channel_norm_backward"] if _plain(gy) else _channel_norm_backward
        gx1, gx2, gw, gb = bwd(gy, x1, x2, weight, mean, rstd, gskip)
        return gx1, (gx2 if x2 is not None else None), gw, gb, None, None


def channel_norm(x, weight, bias, eps: float = 1e-5, x_extra=None, out_bf16: bool = False):
    """ChannelNorm over channels of ``x`` (and, virtually concatenated after them, ``x_extra``).
    ``out_bf16``: a request (see ``_want_bf16_out``) - the output as a bf16 tensor for a pointwise consumer."""
    require_hip(x, x_extra, weight, bias)
    return _channel_norm(x, x_extra, weight, bias, float(eps), _want_bf16_out(x, out_bf16))[0]


def channel_norm_skip(x, weight, bias, eps: float = 1e-5, x_extra=None, out_bf16: bool = False):
    """``(channel_norm(x), x)`` for a block with a residual branch around it (see ``_ChannelNormSkip``)."""
    require_hip(x, x_extra, weight, bias)
    if torch.compiler.is_compiling():
        return _channel_norm(x, x_extra, weight, bias, float(eps))[0], x
    return _ChannelNormSkip.apply(x, x_extra, weight, bias, float(eps), _want_bf16_out(x, out_bf16))


# ---------------------------------------------------------------------------
# a9 GlobalBias map (reference model/blocks.py:188-196)
# ---------------------------------------------------------------------------
@_define("global_bias_map(Tensor A, Tensor U, Tensor V, Tensor? Pw) -> (Tensor, Tensor)")
def _global_bias_map(A, U, V, Pw):
    _f32(A, U, V, Pw)
    A, U, V = A.contiguous(), U.contiguous(), V.contiguous()
    Cin, R = A.shape
    H, W = U.shape[1], V.shape[1]
    if Pw is not None:
        Pw = Pw.contiguous()
        Co = Pw.shape[0]
        m8 = torch.empty(Cin, H, W, dtype=A.dtype, device=A.device)
    else:
        Co, m8 = Cin, A.new_empty(0)
    out = torch.empty(Co, H, W, dtype=A.dtype, device=A.device)
    check(lib.paradis_global_bias_map_fwd(dptr(A), dptr(U), dptr(V), dptr(Pw), dptr(m8) if Pw is not None else None,
                                          dptr(out), Cin, Co, R, H, W, stream_ptr()), "global_bias_map_fwd")
    return out, m8


@_fake("global_bias_map")
def _(A, U, V, Pw):
    Cin, H, W = A.shape[0], U.shape[1], V.shape[1]
    if Pw is not None:
        return A.new_empty(Pw.shape[0], H, W), A.new_empty(Cin, H, W)
    return A.new_empty(Cin, H, W), A.new_empty(0)


@_define("global_bias_map_backward(Tensor gmap, Tensor A, Tensor U, Tensor V, Tensor? Pw, Tensor m8) "
         "-> (Tensor, Tensor, Tensor, Tensor)")
def _global_bias_map_backward(gmap, A, U, V, Pw, m8):
    _f32(gmap)
    A, U, V = A.contiguous(), U.contiguous(), V.contiguous()
    Cin, R = A.shape
    H, W = U.shape[1], V.shape[1]
    gmap = gmap.contiguous()
    Co = gmap.shape[0]
    has_proj = Pw is not None
    gA, gU, gV = torch.empty_like(A), torch.empty_like(U), torch.empty_like(V)
    gPw = torch.empty_like(Pw) if has_proj else A.new_empty(0)
    ws = _ws(lib.paradis_global_bias_map_bwd_ws_bytes(Cin, Co, R, H, W), A.device)
    check(lib.paradis_global_bias_map_bwd(dptr(gmap), dptr(A), dptr(U), dptr(V),
                                          dptr(Pw.contiguous()) if has_proj else None,
                                          dptr(m8) if has_proj else None, dptr(gA), dptr(gU), dptr(gV),
                                          dptr(gPw) if has_proj else None, Cin, Co, R, H, W, dptr(ws),
                                          stream_ptr()), "global_bias_map_bwd")
    return gA, gU, gV, gPw


@_fake("global_bias_map_backward")
def _(gmap, A, U, V, Pw, m8):
    return (A.new_empty(A.shape), U.new_empty(U.shape), V.new_empty(V.shape),
            A.new_empty(Pw.shape) if Pw is not None else A.new_empty(0))


def _gbm_setup(ctx, inputs, output):
    A, U, V, Pw = inputs
    ctx.save_for_backward(A, U, V, Pw, output[1])
    ctx.mark_non_differentiable(output[1])
    ctx.set_materialize_grads(False)


def _gbm_backward(ctx, gmap, gm8=None):
    A, U, V, Pw, m8 = ctx.saved_tensors
    if gmap is None:
        return None, None, None, None
    gA, gU, gV, gPw = _global_bias_map_backward(gmap, A, U, V, Pw, m8)
    return gA, gU, gV, (gPw if Pw is not None else None)


_autograd("global_bias_map", _gbm_setup, _gbm_backward)


def global_bias_map(A, U, V, Pw=None):
    require_hip(A, U, V, Pw)
    return _global_bias_map(A, U, V, Pw)[0]


@_define("global_bias_m8(Tensor A, Tensor U, Tensor V) -> Tensor")
def _global_bias_m8(A, U, V):
    """Un-projected rank-R map m8[Cin,H,W]; the projection to the layer width is applied inside
    the GEMM epilogue (``pointwise(..., bias_proj=(m8, Pw))``) so the [Co,H,W] map is never stored."""
    _f32(A, U, V)
    A, U, V = A.contiguous(), U.contiguous(), V.contiguous()
    Cin, R = A.shape
    H, W = U.shape[1], V.shape[1]
    m8 = torch.empty(Cin, H, W, dtype=A.dtype, device=A.device)
    check(lib.paradis_global_bias_m8_fwd(dptr(A), dptr(U), dptr(V), dptr(m8), Cin, R, H, W, stream_ptr()),
          "global_bias_m8_fwd")
    return m8


@_fake("global_bias_m8")
def _(A, U, V):
    return A.new_empty(A.shape[0], U.shape[1], V.shape[1])


@_define("global_bias_m8_backward(Tensor gm8, Tensor A, Tensor U, Tensor V) -> (Tensor, Tensor, Tensor)")
def _global_bias_m8_backward(gm8, A, U, V):
    _f32(gm8)
    A, U, V = A.contiguous(), U.contiguous(), V.contiguous()
    Cin, R = A.shape
    H, W = U.shape[1], V.shape[1]
    gm8 = gm8.contiguous()
    gA, gU, gV = torch.empty_like(A), torch.empty_like(U), torch.empty_like(V)
    ws = _ws(lib.paradis_global_bias_map_bwd_ws_bytes(Cin, Cin, R, H, W), A.device)
    check(lib.paradis_global_bias_m8_bwd(dptr(gm8), dptr(A), dptr(U), dptr(V), dptr(gA), dptr(gU), dptr(gV),
                                         Cin, R, H, W, dptr(ws), stream_ptr()), "global_bias_m8_bwd")
    return gA, gU, gV


@_fake("global_bias_m8_backward")
def _(gm8, A, U, V):
    return A.new_empty(A.shape), U.new_empty(U.shape), V.new_empty(V.shape)


def _gb8_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _gb8_backward(ctx, gm8):
    return _global_bias_m8_backward(gm8, *ctx.saved_tensors)


_autograd("global_bias_m8", _gb8_setup, _gb8_backward)


def global_bias_m8(A, U, V):
    require_hip(A, U, V)
    return _global_bias_m8(A, U, V)


# ---------------------------------------------------------------------------
# a6 pointwise channel mixing on the matrix cores with fused epilogue
#     y = residual + act(W x + bias + bias_map)
# (reference model/blocks.py:86,110 + :196 + activation + the residual adds of paradis.py:246,253)
# ---------------------------------------------------------------------------
# Arithmetic of the pointwise GEMMs (include/paradis_hip.h, a6), all fp32 in / accumulate / out:
#   "bf16x3" (default) three bf16 terms per operand value (an exact decomposition of the 24-bit significand with
#            fp32's exponent range per element), six products on the bf16 matrix pipe; error vs fp64 not above the
#            f32 MFMA chain's (tests/test_hip_gemm_split.py);
#   "exact"  the f32 MFMA chain (v_mfma_f32_32x32x2_f32);
#   "f16x2"  OPT-IN, not reference-width arithmetic: two f16 terms of the per-TENSOR scaled operands (a block-exponent
#            format: 22 significand bits relative to the tensor's largest magnitude), three products on the f16 pipe.
# PARADIS_GEMM selects ("split" = bf16x3, the name of earlier rounds).  The scheme is an explicit integer argument of
# the ops below (as it is at the C ABI), so a traced graph pins the arithmetic it was traced with.
#   "bf16"   NOT fp32 arithmetic: the reference's bf16-mixed (AMP) training mode (train.py:56, use_amp: true) - operands
#            rounded to bf16, ONE product, fp32 accumulate, results rounded to bf16 where autocast's conv2d rounds them.
#            Selected by ``pointwise`` only inside ``torch.autocast(device_type="cuda", dtype=torch.bfloat16)`` (or by an
#            explicit ``scheme=``); PARADIS_GEMM cannot name it.
GEMM_EXACT, GEMM_BF16, GEMM_F16X2, GEMM_BF16X3 = 0, 1, 2, 3
BF16_STORAGE = os.environ.get("PARADIS_BF16_STORAGE", "1") != "0"   # bf16-mixed scheme: chained activations as bf16 TENSORS (A/B switch)
IO_X16, IO_Y16, IO_Z16, IO_DY16 = 1, 2, 4, 8       # include/paradis_hip.h PARADIS_IO_*: which tensors of a call are bf16
_SCHEMES = {"exact": GEMM_EXACT, "f16x2": GEMM_F16X2, "bf16x3": GEMM_BF16X3, "split": GEMM_BF16X3}
AMAX_PARTIALS = 1024


def _scheme_from_env() -> int:
    name = os.environ.get("PARADIS_GEMM", "bf16x3")
    if name not in _SCHEMES:
        raise ValueError(f"PARADIS_GEMM={name!r}: choose between {'|'.join(_SCHEMES)}")
    return _SCHEMES[name]


GEMM_SCHEME = _scheme_from_env()     # what the Python wrappers pass to the ops when the caller names no scheme


def gemm_scheme_name() -> str:
    return {GEMM_EXACT: "exact", GEMM_BF16: "bf16", GEMM_F16X2: "f16x2", GEMM_BF16X3: "bf16x3"}[GEMM_SCHEME]


def autocast_scheme(default: int) -> int:
    """GEMM arithmetic of a ``pointwise`` call that names none: inside ``torch.autocast("cuda", dtype=torch.bfloat16)`` -
    the reference's ``precision="bf16-mixed"`` (train.py:56; its shipped default, ``use_amp: true``) - the one-product
    bf16 scheme, i.e. what autocast makes of the reference's ``nn.Conv2d`` calls (model/blocks.py:86,110); the fp32
    scheme otherwise.  Everything else on the path (advection, stencils, norms) stays fp32 under autocast, as autocast
    keeps ``grid_sampler``; the scheme travels as an explicit argument of the op, so the backward and a traced graph
    use the arithmetic the forward was called with."""
    if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return GEMM_BF16
    return default


# Split tile images of the weights (split GEMMs).  DEFAULT (round 6): every forward call builds the image from the
# weight it is handed - one launch that also writes the image of W^T when the call is being recorded (the data gradient
# needs it); that second image travels to the backward in the autograd context.  Nothing is cached across calls, so no
# write to a parameter - ``p.data.mul_``, a raw-pointer kernel, a replayed graph - can ever be missed (rounds 1-5
# cached the images keyed on the version counter and an optimiser-step epoch, and asked the user to call
# ``ops.weights_updated()`` after writes neither sees: a silent wrong-answer mode, verdict r5 item 9).  At one forward
# per optimiser step - training with S = 1 - the launches are exactly those of the cache.
# OPT-IN: inside ``with ops.frozen_weights():`` (inference loops, many forwards on fixed weights) the images ARE
# cached across calls, keyed on the parameter object, its data pointer, its autograd version and WEIGHT_EPOCH, which
# ``ops.weights_updated()`` and every optimiser step bump: there the hint is what it says - an optimisation the caller
# asked for by declaring the weights frozen.
WEIGHT_EPOCH = 0
FROZEN_WEIGHTS = False
_IMAGES = {}     # frozen mode only: (id(weight), transpose, scheme) -> (weakref, data_ptr, version, epoch, image)
_TLS = threading.local()


class frozen_weights:
    """``with ops.frozen_weights():`` - the caller declares that no parameter changes inside the block except through
    torch in-place ops (version counter), optimiser steps or writes followed by ``ops.weights_updated()``: weight images
    are reused across forward calls.  Leaving the block drops them."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)

    def __enter__(self):
        global FROZEN_WEIGHTS
        self.prev = FROZEN_WEIGHTS
        FROZEN_WEIGHTS = self.enabled
        return self

    def __exit__(self, *exc):
        global FROZEN_WEIGHTS
        FROZEN_WEIGHTS = self.prev
        if not FROZEN_WEIGHTS:
            _IMAGES.clear()
        return False


def weights_updated() -> None:
    """Invalidate the weight images cached under ``frozen_weights()``.  Outside that mode nothing is cached and the
    call is a no-op kept for callers of rounds 1-5."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def _optimizer_step_hook(optimizer, args, kwargs) -> None:
    weights_updated()


_STEP_HOOK = None


def _ensure_step_hook() -> None:
    """The process-wide optimiser post-hook exists from the moment the first weight image is cached (frozen mode), not
    from import: a process that never enters ``frozen_weights()`` keeps torch.optim untouched."""
    global _STEP_HOOK
    if _STEP_HOOK is None:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        _STEP_HOOK = register_optimizer_step_post_hook(_optimizer_step_hook)


def _drop_images(wid: int) -> None:
    for scheme in (GEMM_BF16, GEMM_F16X2, GEMM_BF16X3):
        _IMAGES.pop((wid, False, scheme), None)
        _IMAGES.pop((wid, True, scheme), None)


def _version_of(t: Tensor) -> int:
    """autograd version counter; inference tensors (created under ``torch.inference_mode()``, the mode Lightning
    runs validation / predict steps in) do not track one and cannot be updated in place outside that mode: 0."""
    return 0 if t.is_inference() else t._version


_PAIRED_SCHEMES = (GEMM_BF16X3, GEMM_BF16)       # schemes whose W and W^T images come from one launch


def _build_images(weight: Tensor, Co: int, Ci: int, transpose: bool, scheme: int, pair: bool):
    """(image, image of the transpose or None): ``pair`` (bf16x3 / bf16-mixed, not ``transpose``) writes both from one launch"""
    w2 = weight.reshape(Co, Ci).contiguous()
    nbytes = lib.paradis_pw_gemm_split_bytes(Ci, Co, scheme) if transpose else \
        lib.paradis_pw_gemm_split_bytes(Co, Ci, scheme)
    out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    if pair and not transpose and scheme in _PAIRED_SCHEMES:
        out_t = torch.empty(lib.paradis_pw_gemm_split_bytes(Ci, Co, scheme), dtype=torch.uint8, device=weight.device)
        check(lib.paradis_pw_gemm_split_weights_pair_scheme(dptr(w2), Co, Ci, scheme, dptr(out), dptr(out_t), stream_ptr()),
              "pw_gemm_split_weights_pair")
        return out, out_t
    check(lib.paradis_pw_gemm_split_weights(dptr(w2), Co, Ci, 1 if transpose else 0, scheme, dptr(out),
                                            stream_ptr()), "pw_gemm_split_weights")
    return out, None


def _split_image(weight: Tensor, Co: int, Ci: int, transpose: bool, scheme: int, want_wt: bool = False):
    """(image of W - or of W^T if ``transpose`` -, image of W^T or None).  ``want_wt`` (forward image only): the caller's
    backward will need the image of W^T - both come from one launch.  An argument of the ``pointwise`` op (the public
    wrapper computes it where grad mode is still visible), not process state: autograd runs backward nodes - and
    checkpoint recomputes their forwards - on its own threads."""
    if not FROZEN_WEIGHTS:
        return _build_images(weight, Co, Ci, transpose, scheme, want_wt)
    key = (id(weight), transpose, scheme)
    ver = _version_of(weight)

    def valid(ent):
        return (ent is not None and ent[0]() is weight and ent[1] == weight.data_ptr() and ent[2] == ver
                and ent[3] == WEIGHT_EPOCH)
    ent = _IMAGES.get(key)
    ent_t = _IMAGES.get((id(weight), True, scheme)) if (want_wt and not transpose) else None
    if valid(ent) and (not (want_wt and not transpose and scheme in _PAIRED_SCHEMES) or valid(ent_t)):
        return ent[4], (ent_t[4] if valid(ent_t) else None)
    out, out_t = _build_images(weight, Co, Ci, transpose, scheme, want_wt)
    if isinstance(weight, torch.nn.Parameter) or weight.is_leaf:
        wid = id(weight)
        _ensure_step_hook()
        try:
            ref = weakref.ref(weight, lambda _r, wid=wid: _drop_images(wid))
            _IMAGES[key] = (ref, weight.data_ptr(), ver, WEIGHT_EPOCH, out)
            if out_t is not None:
                _IMAGES[(wid, True, scheme)] = (ref, weight.data_ptr(), ver, WEIGHT_EPOCH, out_t)
        except TypeError:
            pass
    return out, out_t


def _stash_wt(weight: Tensor, image: Optional[Tensor]) -> None:
    """forward kernel -> its own setup_context (same thread, next Python call): the W^T image of THIS call"""
    _TLS.wt = (weight, image) if image is not None else None


def _take_wt(weight: Tensor) -> Optional[Tensor]:
    ent = getattr(_TLS, "wt", None)
    _TLS.wt = None
    return ent[1] if (ent is not None and ent[0] is weight) else None


@_define("amax_partials(Tensor x) -> Tensor")
def _amax_partials(x):
    """int32[AMAX_PARTIALS]: bit patterns of partial maxima of |x| ([B,C,H,W], channel-sliced views allowed);
    the opt-in f16x2 GEMMs take the maximum of the words as the tensor's largest magnitude.  One read pass."""
    _f32(x)
    x, x_bs = _plane_view(x)
    B, C, H, W = x.shape
    out = torch.empty(AMAX_PARTIALS, dtype=torch.int32, device=x.device)
    check(lib.paradis_amax_partials(dptr(x), B, C * H * W, x_bs, dptr(out), stream_ptr()), "amax_partials")
    return out


@_fake("amax_partials")
def _(x):
    return x.new_empty(AMAX_PARTIALS, dtype=torch.int32)


@_define("pointwise(Tensor x, Tensor weight, Tensor? bias, Tensor? bmap, Tensor? residual, int act, "
         "Tensor? x_pre, int x_act, bool defer_act_grad, Tensor? m8, Tensor? pw, bool save_z, int scheme, "
         "Tensor? gate, bool want_wt_image=False, bool out_bf16=False) -> (Tensor, Tensor, Tensor)", autocast=((0, 6), 12))
def _pointwise(x, weight, bias, bmap, residual, act, x_pre, x_act, defer_act_grad, m8, pw, save_z, scheme, gate,
               want_wt_image=False, out_bf16=False):
    """y = residual + act(W x + bias + bias_map); second output = pre-activation z (empty unless save_z);
    third = amax partials of x (f16x2 scheme; empty otherwise), kept for the weight gradient.
    ``scheme``: GEMM arithmetic (GEMM_EXACT / GEMM_BF16X3 / GEMM_F16X2); the backward uses the same.
    ``gate`` [Co] (with ``residual``): y = residual + sigmoid(gate)[:,None] * (act(...) - residual) - the gated blend
    of reference model/paradis.py:239-243 in the GEMM epilogue (``ops.gated_blend`` without the advected tensor).
    ``want_wt_image``: a hint, no effect on the result - the backward of this call will need the split image of W^T
    (x requires a gradient): both images are then written by one launch and cached.

    Activation-gradient hand-off between two chained ops (GMBlock drives it):
      * ``defer_act_grad`` (producer): the op's backward receives d(pre-activation) directly and
        skips its own act' pass; the consumer reads the returned ``z``.
      * ``x_pre`` / ``x_act`` (consumer): x = act(x_pre) was produced by such an op; the consumer's
        dgrad multiplies by act'(x_pre) in the GEMM epilogue, so what it returns as the gradient of
        ``x`` already is the producer's d(pre-activation).
    """
    _f32(weight, bias, bmap, residual, m8, pw, gate)
    _f32_or_bf16(scheme, x)
    x, x_bs = _plane_view(x)
    B, Ci, H, W = x.shape
    Co = weight.shape[0]
    P = H * W
    assert weight.numel() == Co * Ci, "weight/in-channel mismatch"
    x16 = _is16(x)
    if x16 or out_bf16:
        # bf16-STORED activations (bf16-mixed scheme only; the public wrapper decides): ``out_bf16`` writes y and z as
        # bf16 tensors - what autocast's conv2d returns in the reference (model/blocks.py:86,110)
        if scheme != GEMM_BF16 or (out_bf16 and residual is not None):
            raise RuntimeError("bf16-stored tensors need the bf16-mixed GEMM scheme (and a bf16 output no residual)")
        if x16 and (P % 8 or x_bs % 8 or x.data_ptr() % 16):
            raise RuntimeError("a bf16-stored GEMM input needs H W % 8 == 0 and 16-byte aligned planes")
    if gate is not None:
        assert residual is not None and gate.numel() == Co, "a gate needs the residual it blends with, one value per channel"
        gate = gate.reshape(Co).contiguous()
    res_bs = 0
    if residual is not None:
        residual, res_bs = _plane_view(residual)
    if bmap is not None:
        bmap = bmap.contiguous()
    y = torch.empty(B, Co, H, W, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    cin, pwt = 0, None
    if pw is not None:
        m8, pw = m8.contiguous(), pw.contiguous()
        cin = pw.shape[1]
        assert pw.shape[0] == Co and m8.shape == (cin, H, W) and bmap is None
        assert Co % 4 == 0 and cin <= 16, "fused GlobalBias projection needs Co % 4 == 0 and <= 16 bias channels"
        pwt = pw.t().contiguous()   # [cin, Co]: four consecutive output rows per 16-byte load
    z = torch.empty_like(y) if (save_z and act != 0) else y.new_empty(0)
    w2 = w2t = wsp = None
    x_amax = x.new_empty(0, dtype=torch.int32)
    _stash_wt(weight, None)
    if scheme != GEMM_EXACT:
        wsp, wtsp_out = _split_image(weight, Co, Ci, False, scheme, bool(want_wt_image))   # split planes in tile order
        _stash_wt(weight, wtsp_out)
        if scheme == GEMM_F16X2:
            x_amax = torch.empty(AMAX_PARTIALS, dtype=torch.int32, device=x.device)
            check(lib.paradis_amax_partials(dptr(x), B, Ci * P, x_bs, dptr(x_amax), stream_ptr()), "amax_partials")
    else:
        w2 = weight.reshape(Co, Ci).contiguous()
        if Ci % 16 == 0 and Co % 4 == 0 and Co * Ci >= 4096:
            # [Ci,Co] copy of the weights: makes the A operand row-contiguous for the LDS-DMA kernel
            w2t = torch.empty(Ci, Co, dtype=torch.float32, device=x.device)
            check(lib.paradis_transpose(dptr(w2), dptr(w2t), Co, Ci, stream_ptr()), "transpose")
    if w2 is None:
        w2 = weight.reshape(Co, Ci)
        if not w2.is_contiguous():
            w2 = w2.contiguous()
    head = (dptr(w2), dptr(w2t), dptr(wsp), scheme, dptr(x_amax) if x_amax.numel() else None, dptr(x), dptr(bias),
            dptr(bmap), dptr(m8) if cin else None, dptr(pwt) if cin else None, cin, dptr(residual))
    tail = (dptr(y), dptr(z) if z.numel() else None, B, Co, Ci, P, x_bs, res_bs, Co * P, act, stream_ptr())
    if x16 or out_bf16:
        _lib.call("pw_gemm_fwd", 2.0 * B * Co * Ci * P, dptr(wsp), dptr(x), dptr(bias), dptr(bmap),
                  dptr(m8) if cin else None, dptr(pwt) if cin else None, cin, dptr(residual), dptr(gate), dptr(y),
                  dptr(z) if z.numel() else None, B, Co, Ci, P, x_bs, res_bs, Co * P, act,
                  (IO_X16 if x16 else 0) | (IO_Y16 if out_bf16 else 0), stream_ptr(), symbol="pw_gemm_fwd16")
        return y, z, x_amax
    if gate is None:
        _lib.call("pw_gemm_fwd", 2.0 * B * Co * Ci * P, *head, *tail)
    else:
        _lib.call("pw_gemm_fwd", 2.0 * B * Co * Ci * P, *head, dptr(gate), *tail, symbol="pw_gemm_fwd_gated")
    return y, z, x_amax


@_fake("pointwise")
def _(x, weight, bias, bmap, residual, act, x_pre, x_act, defer_act_grad, m8, pw, save_z, scheme, gate,
      want_wt_image=False, out_bf16=False):
    B, _, H, W = x.shape
    y = x.new_empty(B, weight.shape[0], H, W, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    return (y, (y.new_empty(y.shape) if (save_z and act != 0) else y.new_empty(0)),
            x.new_empty(AMAX_PARTIALS if scheme == GEMM_F16X2 else 0, dtype=torch.int32))


@_define("act_backward(Tensor gy, Tensor z, int act, bool out_bf16=False) -> Tensor")
def _act_backward(gy, z, act, out_bf16=False):
    """``out_bf16`` (bf16-mixed scheme): d(pre-activation) as a bf16 tensor - the operand both gradient GEMMs of the layer
    would round to bf16 on load (same values; the reference's autocast backward holds this tensor in bf16 as well)"""
    _f32(gy, z)
    gy, z = gy.contiguous(), z.contiguous()
    if out_bf16 and gy.numel() % 4 == 0:
        dz = torch.empty(gy.shape, dtype=torch.bfloat16, device=gy.device)
        check(lib.paradis_act_bwd16(dptr(gy), dptr(z), dptr(dz), gy.numel(), act, stream_ptr()), "act_bwd16")
        return dz
    dz = torch.empty_like(gy)
    check(lib.paradis_act_bwd(dptr(gy), dptr(z), dptr(dz), gy.numel(), act, stream_ptr()), "act_bwd")
    return dz


@_fake("act_backward")
def _(gy, z, act, out_bf16=False):
    return gy.new_empty(gy.shape, dtype=torch.bfloat16 if (out_bf16 and gy.numel() % 4 == 0) else gy.dtype)


@_define("pw_gemm_dgrad(Tensor dz, Tensor weight, Tensor? zmul, int x_act, Tensor? dz_amax, int scheme, "
         "Tensor? wt_image=None, bool out_bf16=False) -> Tensor", autocast=((0, 2), 5))
def _pw_gemm_dgrad(dz, weight, zmul, x_act, dz_amax, scheme, wt_image=None, out_bf16=False):
    """gx = W^T dz (* act'(zmul) when the producing layer deferred its activation gradient).
    dz_amax: amax partials of dz (f16x2 scheme; computed here when missing).
    wt_image: the split image of W^T the recorded forward wrote next to W's (uint8, bf16x3); built here from ``weight``
    when missing (traced graphs, the one-plane schemes, direct calls)."""
    _f32(weight)
    _f32_or_bf16(scheme, dz, zmul)
    dz = dz.contiguous()
    B, Co, H, W = dz.shape
    Ci = weight.numel() // Co
    P = H * W
    gx = torch.empty(B, Ci, H, W, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dz.device)
    w2 = weight.reshape(Co, Ci)
    if not w2.is_contiguous():
        w2 = w2.contiguous()
    wtsp = None
    if scheme != GEMM_EXACT:
        nbytes = lib.paradis_pw_gemm_split_bytes(Ci, Co, scheme)
        if wt_image is not None and wt_image.dtype == torch.uint8 and wt_image.numel() == nbytes:
            wtsp = wt_image
        else:
            wtsp = _split_image(weight, Co, Ci, True, scheme)[0]
    if scheme == GEMM_F16X2 and dz_amax is None:
        dz_amax = _amax_partials(dz)
    if zmul is not None:
        zmul = zmul.contiguous()
    io = (IO_X16 if _is16(dz) else 0) | (IO_Y16 if out_bf16 else 0) | (IO_Z16 if (x_act != 0 and _is16(zmul)) else 0)
    if io:
        if scheme != GEMM_BF16 or (_is16(dz) and (P % 8 or dz.data_ptr() % 16)):
            raise RuntimeError("bf16-stored gradients need the bf16-mixed GEMM scheme, H W % 8 == 0 and aligned planes")
        _lib.call("pw_gemm_dgrad", 2.0 * B * Co * Ci * P, dptr(wtsp), dptr(dz), dptr(zmul) if x_act != 0 else None,
                  dptr(gx), B, Co, Ci, P, Co * P, Ci * P, Ci * P, x_act, io, stream_ptr(), symbol="pw_gemm_dgrad16")
        return gx
    _lib.call("pw_gemm_dgrad", 2.0 * B * Co * Ci * P, dptr(w2), dptr(wtsp), scheme,
              dptr(dz_amax) if scheme == GEMM_F16X2 else None, dptr(dz),
              dptr(zmul) if x_act != 0 else None, None, dptr(gx), B, Co, Ci, P, Co * P, Ci * P, 0, Ci * P,
              x_act, stream_ptr())
    return gx


@_fake("pw_gemm_dgrad")
def _(dz, weight, zmul, x_act, dz_amax, scheme, wt_image=None, out_bf16=False):
    B, Co, H, W = dz.shape
    return dz.new_empty(B, weight.numel() // Co, H, W, dtype=torch.bfloat16 if out_bf16 else torch.float32)


@_define("pw_gemm_wgrad(Tensor dz, Tensor x, bool want_bias, Tensor? dz_amax, Tensor? x_amax, int scheme) "
         "-> (Tensor, Tensor)", autocast=((0, 1), 5))
def _pw_gemm_wgrad(dz, x, want_bias, dz_amax, x_amax, scheme):
    """gW[Co,Ci] = sum over samples and points of dz x^T; the bias gradient (row sums of dz) falls out
    of the same pass.  dz_amax / x_amax: amax partials (f16x2 scheme; computed here when missing)."""
    _f32_or_bf16(scheme, dz, x)
    dz = dz.contiguous()
    x, x_bs = _plane_view(x)
    B, Co, H, W = dz.shape
    Ci, P = x.shape[1], H * W
    gw = torch.empty(Co, Ci, dtype=torch.float32, device=dz.device)
    gb = torch.empty(Co if want_bias else 0, dtype=torch.float32, device=dz.device)
    ws = _ws(lib.paradis_pw_gemm_wgrad_ws_bytes(B, Co, Ci, P), dz.device)
    io = (IO_DY16 if _is16(dz) else 0) | (IO_X16 if _is16(x) else 0)
    if io:
        if P % 16 or dz.data_ptr() % 16 or x.data_ptr() % 16 or x_bs % 8:
            raise RuntimeError("bf16-stored GEMM operands need H W % 16 == 0 and 16-byte aligned planes")
        _lib.call("pw_gemm_wgrad", 2.0 * B * Co * Ci * P, dptr(dz), dptr(x), dptr(gw), dptr(gb) if want_bias else None,
                  B, Co, Ci, P, Co * P, x_bs, io, dptr(ws), stream_ptr(), symbol="pw_gemm_wgrad16")
        return gw, gb
    f16 = scheme == GEMM_F16X2
    if f16 and (dz_amax is None or dz_amax.numel() == 0):
        dz_amax = _amax_partials(dz)
    if f16 and (x_amax is None or x_amax.numel() == 0):
        x_amax = _amax_partials(x)
    _lib.call("pw_gemm_wgrad", 2.0 * B * Co * Ci * P, dptr(dz), dptr(x), dptr(gw), dptr(gb) if want_bias else None,
              B, Co, Ci, P, Co * P, x_bs, scheme, dptr(dz_amax) if f16 else None,
              dptr(x_amax) if f16 else None, dptr(ws), stream_ptr())
    return gw, gb


@_fake("pw_gemm_wgrad")
def _(dz, x, want_bias, dz_amax, x_amax, scheme):
    Co, Ci = dz.shape[1], x.shape[1]
    return dz.new_empty(Co, Ci, dtype=torch.float32), dz.new_empty(Co if want_bias else 0, dtype=torch.float32)


@_define("bias_grads(Tensor dz, bool want_bias, bool want_map) -> (Tensor, Tensor)", autocast=((0,), None))
def _bias_grads(dz, want_bias, want_map):
    """gb[Co] = sum over (B,H,W), gmap[Co,H,W] = sum over B of dz (fp32 sums; dz fp32, or bf16-stored in the bf16-mixed mode)."""
    _f32_or_bf16(GEMM_BF16, dz)
    dz = dz.contiguous()
    B, Co, H, W = dz.shape
    gb = torch.empty(Co if want_bias else 0, dtype=torch.float32, device=dz.device)
    gmap = torch.empty((Co, H, W) if want_map else (0,), dtype=torch.float32, device=dz.device)
    if _is16(dz):
        check(lib.paradis_bias_grads16(dptr(dz), dptr(gmap) if want_map else None, dptr(gb) if want_bias else None, B,
                                       Co, H * W, Co * H * W, stream_ptr()), "bias_grads16")
        return gb, gmap
    check(lib.paradis_bias_grads(dptr(dz), dptr(gmap) if want_map else None, dptr(gb) if want_bias else None, B,
                                 Co, H * W, Co * H * W, stream_ptr()), "bias_grads")
    return gb, gmap


@_fake("bias_grads")
def _(dz, want_bias, want_map):
    B, Co, H, W = dz.shape
    return (dz.new_empty(Co if want_bias else 0, dtype=torch.float32),
            dz.new_empty((Co, H, W) if want_map else (0,), dtype=torch.float32))


@_define("global_bias_proj_backward(Tensor gmap, Tensor m8, Tensor pw) -> (Tensor, Tensor)")
def _global_bias_proj_backward(gmap, m8, pw):
    """adjoint of the projection fused into the GEMM epilogue: gmap[Co,H,W] -> (gPw[Co,cin], gm8[cin,H,W])"""
    _f32(gmap, m8, pw)
    gmap, m8, pw = gmap.contiguous(), m8.contiguous(), pw.contiguous()
    Co, cin = pw.shape
    gpw, gm8 = torch.empty_like(pw), torch.empty_like(m8)
    check(lib.paradis_global_bias_proj_bwd(dptr(gmap), dptr(m8), dptr(pw), dptr(gpw), dptr(gm8), cin, Co,
                                           m8.shape[1] * m8.shape[2], stream_ptr()), "global_bias_proj_bwd")
    return gpw, gm8


@_fake("global_bias_proj_backward")
def _(gmap, m8, pw):
    return pw.new_empty(pw.shape), m8.new_empty(m8.shape)


def _pw_setup(ctx, inputs, output):
    x, weight, bias, bmap, residual, act, x_pre, x_act, defer, m8, pw, save_z, scheme, gate, _want_wt, out16 = inputs
    y, z, x_amax = output
    gated = gate is not None
    # (a gated epilogue blends with its residual: the backward needs both ends of the blend - the residual and the
    #  output, which the next block keeps alive anyway - and the gate)
    ctx.save_for_backward(x, weight, z, x_pre, m8, pw, x_amax, *((residual, y, gate) if gated else (None, None, None)))
    ctx.meta = (act, bias is not None, bmap is not None, residual is not None,
                x_act if x_pre is not None else 0, bool(defer), scheme)
    ctx.out16 = bool(out16)
    ctx.wt_image = _take_wt(weight)      # the W^T image this very forward call wrote (None: the dgrad builds its own)
    # z carries no gradient; without this autograd would materialise a full-size zero tensor for it
    ctx.mark_non_differentiable(z, x_amax)
    ctx.set_materialize_grads(False)


# ---- weight gradients on a side stream (round 6) -------------------------------------------------------------
# The weight-gradient GEMM of a pointwise layer depends on (dz, saved x) only; nothing of the dgrad chain waits for it
# (reference model/paradis.py:228-254: per-layer chain).  Issued on a second HIP stream it runs UNDER the memory- /
# VALU-bound kernels of the blocks that follow in backward (ChannelNorm, stencil, advection, blend) instead of in
# series with them.  Ordering contract:
#   * the side stream waits for an event recorded on the launching stream right after dz exists;
#   * dz / x / amax tensors are kept referenced until the launching stream has waited for the GEMM's completion event
#     (lagged by ``WgradSide.LAG`` launches, so that wait never stalls), so the caching allocator cannot hand their
#     blocks to later launching-stream kernels while the side stream still reads them - no ``record_stream`` (which a
#     graph capture would turn into memory held to the end of the capture);
#   * gW (and the fused bias gradient) are allocated ON the side stream; whoever consumes them on the launching stream
#     - autograd's AccumulateGrad (a pointer move while ``.grad`` is None), the optimiser, DDP's bucket copy - must be
#     ordered after the side stream: ``WgradSide.join()`` runs as an autograd-engine callback at the end of every
#     backward pass that used the side stream, and ``DelayedGrads`` (model/paradis.py) delivers the gradients of one
#     ADR layer to their AccumulateGrad nodes - hence to DDP's reducer hooks - one layer late, after a join of exactly
#     those launches.
class WgradSide:
    LAG = 6                    # launches kept in flight before the launching stream waits for the oldest
    enabled = os.environ.get("PARADIS_WGRAD_STREAM", "0") == "1"
    PRIORITY = int(os.environ.get("PARADIS_WGRAD_STREAM_PRIORITY", "0"))    # torch.cuda.Stream priority of the side stream
    _streams = {}              # device index -> torch.cuda.Stream
    _pending = []              # [(event, refs)] oldest first
    _callback_queued = False

    @classmethod
    def stream(cls, device) -> "torch.cuda.Stream":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = cls._streams.get(idx)
        if st is None:
            st = cls._streams[idx] = torch.cuda.Stream(device=idx, priority=cls.PRIORITY)
        return st

    @classmethod
    def launch(cls, fn, dz, x, *rest):
        """run ``fn(dz, x, *rest)`` on the side stream, ordered after everything already enqueued on the current one"""
        side = cls.stream(dz.device)
        ready = torch.cuda.Event()
        ready.record()
        side.wait_event(ready)
        with torch.cuda.stream(side):
            out = fn(dz, x, *rest)
            done = torch.cuda.Event()
            done.record()
        cur = torch.cuda.current_stream()
        cls._pending.append((done, cur, (dz, x, rest)))
        while len(cls._pending) > cls.LAG:
            ev, st, _refs = cls._pending.pop(0)
            st.wait_event(ev)
        if not cls._callback_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(cls.join)
                cls._callback_queued = True
            except RuntimeError:       # not inside a backward pass (a direct call of the backward op): join now
                cls.join()
        return out

    @classmethod
    def join(cls) -> None:
        """the current stream waits for every side-stream launch issued so far; references are dropped"""
        cls._callback_queued = False
        # (the engine runs its callbacks on a worker thread whose current stream is not the launching one: every entry
        #  remembers the stream it was launched from)
        for ev, st, _refs in cls._pending:
            st.wait_event(ev)
        cls._pending.clear()


def _pw_backward(ctx, gy, gz=None, gamax=None, raw=False):
    """``raw``: called from ``_PointwiseEager.backward`` - an eager, un-traced backward: the HIP kernels' Python functions
    are called directly instead of through the dispatcher (~250 op calls per training step at ~12 us each of dispatcher
    -> autograd key -> autocast key -> Python kernel; the arithmetic is the same code)."""
    saved = ctx.saved_tensors        # once: non-reentrant checkpointing unpacks (recomputes) on this access
    if raw is None:
        raw = gy is not None and _plain(gy) and _all_plain(saved)
    K = RAW if raw else OPS          # name -> the kernel's Python function / the OpOverload
    x, weight, z, x_pre, m8, pw, x_amax, res_saved, y_saved, gate = saved
    act, has_bias, has_map, has_res, x_act, deferred, scheme = ctx.meta
    need = ctx.needs_input_grad
    if gy is None:
        return (None,) * 16
    # the cotangent of a bf16-stored output arrives as bf16 (autograd keeps a gradient in its tensor's dtype); anything
    # else - a cast a caller put in between, an fp32 cotangent for a bf16 output - is brought to the dtype the kernels of
    # this node expect
    want = torch.bfloat16 if getattr(ctx, "out16", False) else torch.float32
    if gy.dtype != want:
        gy = gy.to(want)
    has_proj = pw is not None
    ggate = None
    if gate is not None:
        # gradient of the blend: d residual, d (activated GEMM output) - what the rest of this backward continues
        # with - and d gate, in one pass over (gy, residual, y)
        gres, gy, ggate = K["gated_blend_backward_out"](gy, res_saved, y_saved, gate)
        ggate = ggate.reshape(gate.shape)
    else:
        gres = gy if has_res else None
    if act != 0 and not deferred:
        # (bf16-mixed scheme: dz leaves the activation-gradient pass as a bf16 tensor - what its consumers, the two gradient
        #  GEMMs, round it to anyway - where their bf16-operand layout rules hold)
        dz16 = (scheme == GEMM_BF16 and BF16_STORAGE and gy.dtype == torch.float32 and (gy.shape[-2] * gy.shape[-1]) % 16 == 0)
        dz = K["act_backward"](gy, z, act, dz16)
    else:
        dz = gy          # no activation, or the consumer already applied act'(z) (deferred)
    gx = gw = gb = gmap = gm8 = gpw = None
    # f16x2: one read pass over dz serves both of its GEMMs
    dz_amax = K["amax_partials"](dz) if (scheme == GEMM_F16X2 and (need[0] or need[1])) else None
    if need[0]:
        # (a bf16-stored input gets its gradient as bf16: with the activation-gradient hand-off that IS the producing
        #  layer's d(pre-activation), rounded to bf16 where the reference's autocast backward rounds it)
        gx = K["pw_gemm_dgrad"](dz, weight, x_pre if x_act != 0 else None, x_act, dz_amax, scheme,
                                getattr(ctx, "wt_image", None), x.dtype == torch.bfloat16)
    want_b = has_bias and need[2]
    want_p = has_proj and (need[9] or need[10])
    want_m = (has_map and need[3]) or want_p
    if need[1]:
        fused_b = want_b and not want_m     # bias gradient = row sums of dz: fused into the wgrad GEMM
        if raw and WgradSide.enabled:
            gw, gbf = WgradSide.launch(K["pw_gemm_wgrad"], dz, x, fused_b, dz_amax, x_amax, scheme)
        else:
            gw, gbf = K["pw_gemm_wgrad"](dz, x, fused_b, dz_amax, x_amax, scheme)
        gw = gw.reshape(weight.shape)
        if fused_b:
            gb, want_b = gbf, False
    if want_b or want_m:
        gb2, gmap = K["bias_grads"](dz, want_b, want_m)
        if want_b:
            gb = gb2
        if not want_m:
            gmap = None
    if want_p:   # adjoint of the fused projection: gmap -> (gPw, gm8); the full map only lives here
        gpw, gm8 = K["global_bias_proj_backward"](gmap, m8, pw)
        gmap = None
    return gx, gw, gb, gmap, gres, None, None, None, None, gm8, gpw, None, None, ggate, None, None


_autograd("pointwise", _pw_setup, _pw_backward)


class _PointwiseEager(torch.autograd.Function):
    """``paradis::pointwise`` with its registered setup / backward, for eager recording calls (see ``pointwise``)."""

    @staticmethod
    def forward(ctx, *args):
        # (inside autocast the op's autocast rule widens bf16 / fp16 inputs: through the dispatcher; otherwise straight
        #  to the kernel's Python function - same checks, same launch)
        # (dtypes were settled by ``pointwise``: fp32 everywhere except the bf16-stored activations of the bf16-mixed
        #  scheme - the op has no blanket autocast rule)
        out = _eager_forward("pointwise", _pointwise, *args)
        _pw_setup(ctx, args, out)
        return out

    @staticmethod
    def backward(ctx, gy, gz=None, gamax=None):
        return _pw_backward(ctx, gy, gz, gamax, raw=None)   # None: decided on the unpacked tensors


def pointwise(x, weight, bias=None, bias_map=None, residual=None, act=None, x_pre=None, x_act=None,
              defer_act_grad=False, bias_proj=None, scheme: Optional[int] = None, gate=None, out_bf16: bool = False):
    """y = residual + act(weight . x + bias[:,None] + bias_map); weight [Co,Ci] or [Co,Ci,1,1].

    ``bias_proj=(m8[Cin,H,W], Pw[Co,Cin])`` adds the projected low-rank GlobalBias map inside the GEMM
    epilogue instead of a materialised ``bias_map``.
    ``scheme``: GEMM arithmetic, default ``ops.GEMM_SCHEME`` (read when the call is made / traced); inside
    ``torch.autocast("cuda", dtype=torch.bfloat16)`` the default is ``GEMM_BF16`` (``autocast_scheme``).
    ``defer_act_grad=True`` returns ``(y, z)`` and expects the consumer to be another ``pointwise``
    called with ``x_pre=z, x_act=act`` (see the op docstring); only valid when ``y`` has no other use.
    ``gate`` [Co] (needs ``residual``): y = residual + sigmoid(gate) * (act(...) - residual), i.e.
    ``gated_blend(residual, pointwise(...), gate)`` without the intermediate tensor (bit-identical).
    ``out_bf16``: a REQUEST, honoured in the bf16-mixed scheme only (and only where the kernels' layout rules hold):
    y (and z) come back as ``torch.bfloat16`` tensors - what the reference's conv2d returns under autocast
    (model/blocks.py:86,110 with train.py:56) - for a consumer that is another ``pointwise``; the values are those of
    the fp32-stored result bit for bit.  A bf16 ``x`` / ``x_pre`` is consumed as stored in that scheme and widened to
    fp32 in every other."""
    if defer_act_grad and (act is None or residual is not None):
        raise ValueError("defer_act_grad needs an activation and no residual")
    if gate is not None and residual is None:
        raise ValueError("a gate needs the residual it blends with")
    m8, pw = bias_proj if bias_proj is not None else (None, None)
    require_hip(x, weight, bias, bias_map, residual, x_pre, m8, pw, gate)
    sch = autocast_scheme(GEMM_SCHEME) if scheme is None else int(scheme)
    # dtypes (the op carries no blanket autocast rule): everything fp32, except that the bf16-mixed scheme consumes a
    # bf16-stored activation (and its pre-activation) as it is where the layout allows the 16-byte DMA
    P = x.shape[-2] * x.shape[-1]
    ok16 = sch == GEMM_BF16 and P % 16 == 0 and BF16_STORAGE
    if x.dtype != torch.float32 and not (ok16 and x.dtype == torch.bfloat16):
        x = x.float()
    if x_pre is not None and x_pre.dtype != torch.float32 and not (ok16 and x_pre.dtype == torch.bfloat16):
        x_pre = x_pre.float()
    weight, bias, bias_map, residual, m8, pw, gate = (
        t.float() if (t is not None and t.dtype != torch.float32) else t
        for t in (weight, bias, bias_map, residual, m8, pw, gate))
    # a bf16 output only where its gradient path exists: the consumer is a pointwise layer taking over the activation
    # gradient (recording), or nothing is recorded
    out16 = bool(out_bf16) and ok16 and residual is None and (bool(defer_act_grad) or not torch.is_grad_enabled())
    code = ACT_CODES[act]
    save_z = bool(defer_act_grad)
    if code != 0 and not save_z and torch.is_grad_enabled():
        save_z = any(t is not None and t.requires_grad for t in (x, weight, bias, bias_map, m8, pw))
    # (grad mode is off inside the op's own forward: the hint is computed here and travels as an argument)
    eager = not torch.compiler.is_compiling()
    want_wt = eager and torch.is_grad_enabled() and (x.requires_grad or (x_pre is not None and x_pre.requires_grad))
    args = (x, weight, bias, bias_map, residual, code, x_pre, ACT_CODES[x_act], bool(defer_act_grad), m8, pw,
            save_z, sch, gate, bool(want_wt), out16)
    # eager + recording: the same forward / setup / backward through a plain autograd.Function - torch.library's generic
    # autograd wrapper spends ~30 us per call of this 15-argument op rebuilding the schema's argument list
    # (torch/_library/utils.py fill_defaults; tools/host_profile.py: 78 calls per step).  Traced graphs keep the op.
    y, z, _ = _PointwiseEager.apply(*args) if (eager and torch.is_grad_enabled()) else _pointwise(*args)
    return (y, z) if defer_act_grad else y


# ---------------------------------------------------------------------------
# elementwise glue
# ---------------------------------------------------------------------------
@_define("activation(Tensor x, int act) -> Tensor")
def _activation(x, act):
    _f32(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib.paradis_act_fwd(dptr(x), dptr(y), x.numel(), act, stream_ptr()), "act_fwd")
    return y


@_fake("activation")
def _(x, act):
    return x.new_empty(x.shape)


def _act_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.act = inputs[1]


def _act_bwd(ctx, gy):
    return _act_backward(gy, ctx.saved_tensors[0], ctx.act), None


_autograd("activation", _act_setup, _act_bwd)


def activation(x, name: str):
    if name not in ("SiLU", "GELU"):
        raise ValueError(f"Unknown activation_fn '{name}'. Allowed: ['SiLU', 'GELU']")
    require_hip(x)
    return _activation(x, ACT_CODES[name])


@_define("gated_blend(Tensor h, Tensor adv, Tensor alpha) -> Tensor")
def _gated_blend(h, adv, alpha):
    _f32(h, adv, alpha)
    h, adv, alpha = h.contiguous(), adv.contiguous(), alpha.contiguous()
    B, C, H, W = h.shape
    out = torch.empty_like(h)
    check(lib.paradis_gated_blend_fwd(dptr(h), dptr(adv), dptr(alpha), dptr(out), B, C, H * W, stream_ptr()),
          "gated_blend_fwd")
    return out


@_fake("gated_blend")
def _(h, adv, alpha):
    return h.new_empty(h.shape)


@_define("gated_blend_backward(Tensor gout, Tensor h, Tensor adv, Tensor alpha) -> (Tensor, Tensor, Tensor)")
def _gated_blend_backward(gout, h, adv, alpha):
    _f32(gout)
    h, adv, alpha, gout = h.contiguous(), adv.contiguous(), alpha.contiguous(), gout.contiguous()
    B, C, H, W = h.shape
    gh, gadv, galpha = torch.empty_like(h), torch.empty_like(h), torch.empty_like(alpha)
    ws = _ws(lib.paradis_gated_blend_bwd_ws_bytes(B, C, H * W), h.device)
    check(lib.paradis_gated_blend_bwd(dptr(gout), dptr(h), dptr(adv), dptr(alpha), dptr(gh), dptr(gadv),
                                      dptr(galpha), B, C, H * W, dptr(ws), stream_ptr()), "gated_blend_bwd")
    return gh, gadv, galpha


@_fake("gated_blend_backward")
def _(gout, h, adv, alpha):
    return h.new_empty(h.shape), h.new_empty(h.shape), alpha.new_empty(alpha.shape)


@_define("gated_blend_backward_out(Tensor gout, Tensor h, Tensor out, Tensor alpha) -> (Tensor, Tensor, Tensor)")
def _gated_blend_backward_out(gout, h, out, alpha):
    """(gh, gadv, galpha) of ``out = h + sigmoid(alpha) (adv - h)`` from the blended output (the gated GEMM epilogue
    never materialises adv): galpha = (1 - sigmoid) sum gout (out - h)."""
    _f32(gout, h, out, alpha)
    gout, h, out = gout.contiguous(), h.contiguous(), out.contiguous()
    B, C, H, W = h.shape
    alpha = alpha.reshape(C).contiguous()
    gh, gadv = torch.empty_like(h), torch.empty_like(h)
    galpha = torch.empty(C, dtype=h.dtype, device=h.device)
    ws = _ws(lib.paradis_gated_blend_bwd_ws_bytes(B, C, H * W), h.device)
    check(lib.paradis_gated_blend_bwd_out(dptr(gout), dptr(h), dptr(out), dptr(alpha), dptr(gh), dptr(gadv),
                                          dptr(galpha), B, C, H * W, dptr(ws), stream_ptr()), "gated_blend_bwd_out")
    return gh, gadv, galpha


@_fake("gated_blend_backward_out")
def _(gout, h, out, alpha):
    return h.new_empty(h.shape), h.new_empty(h.shape), alpha.new_empty(alpha.numel())


def _blend_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _blend_backward(ctx, gout):
    return _gated_blend_backward(gout, *ctx.saved_tensors)


_autograd("gated_blend", _blend_setup, _blend_backward)


def gated_blend(h, adv, alpha):
    """h + sigmoid(alpha)[None,:,None,None] * (adv - h)   (reference model/paradis.py:239-243)"""
    require_hip(h, adv, alpha)
    return _gated_blend(h, adv, alpha)


@_define("add(Tensor a, Tensor b) -> Tensor")
def _add(a, b):
    _f32(a, b)
    assert a.shape == b.shape
    a, b = a.contiguous(), b.contiguous()
    y = torch.empty_like(a)
    check(lib.paradis_add(dptr(a), dptr(b), dptr(y), a.numel(), stream_ptr()), "add")
    return y


@_fake("add")
def _(a, b):
    return a.new_empty(a.shape)


_autograd("add", lambda ctx, inputs, output: None, lambda ctx, gy: (gy, gy))


def add(a, b):
    require_hip(a, b)
    return _add(a, b)


@_define("add_bias_map(Tensor x, Tensor bmap) -> Tensor")
def _add_bias_map(x, bmap):
    _f32(x, bmap)
    x, bmap = x.contiguous(), bmap.contiguous()
    assert x.shape[1:] == bmap.shape
    y = torch.empty_like(x)
    check(lib.paradis_add_bcast(dptr(x), dptr(bmap), dptr(y), bmap.numel(), x.shape[0], stream_ptr()), "add_bcast")
    return y


@_fake("add_bias_map")
def _(x, bmap):
    return x.new_empty(x.shape)


def _abm_backward(ctx, gy):
    gmap = None
    if ctx.needs_input_grad[1]:
        gmap = _bias_grads(gy, False, True)[1]
    return gy, gmap


_autograd("add_bias_map", lambda ctx, inputs, output: None, _abm_backward)


def add_bias_map(x, bmap):
    """x[B,C,H,W] + bmap[C,H,W] (standalone GlobalBias, reference model/blocks.py:196)"""
    require_hip(x, bmap)
    return _add_bias_map(x, bmap)


# ---------------------------------------------------------------------------
# rows f1-f2: loss and rollout glue
# ---------------------------------------------------------------------------
@_define("paradis_loss(Tensor pred, Tensor target, Tensor wf, Tensor? wl, int kind, float delta, bool want_grad) "
         "-> (Tensor, Tensor)")
def _paradis_loss(pred, target, wf, wl, kind, delta, want_grad):
    """mean(wf[c] * wl[h] * l(pred - target)) and, in the same pass, its derivative w.r.t. pred."""
    _f32(pred, target, wf, wl)
    pred, target = pred.contiguous(), target.contiguous()
    B, C, H, W = pred.shape
    assert target.shape == pred.shape and wf.numel() == C and (wl is None or wl.numel() == H)
    loss = torch.empty((), dtype=pred.dtype, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else pred.new_empty(0)
    partial = torch.empty(lib.paradis_loss_blocks(pred.numel()), dtype=pred.dtype, device=pred.device)
    check(lib.paradis_loss_fwd_bwd(dptr(pred), dptr(target), dptr(wf), dptr(wl), dptr(loss),
                                   dptr(grad) if want_grad else None, dptr(partial), B, C, H, W, kind, delta,
                                   stream_ptr()), "loss_fwd_bwd")
    return loss, grad


@_fake("paradis_loss")
def _(pred, target, wf, wl, kind, delta, want_grad):
    return pred.new_empty(()), (pred.new_empty(pred.shape) if want_grad else pred.new_empty(0))


@_define("scale(Tensor x, Tensor s) -> Tensor")
def _scale(x, s):
    """x * s for a one-element device tensor s (no host sync)"""
    _f32(x, s)
    x, s = x.contiguous(), s.contiguous()
    out = torch.empty_like(x)
    check(lib.paradis_scale(dptr(x), dptr(s), dptr(out), x.numel(), stream_ptr()), "scale")
    return out


@_fake("scale")
def _(x, s):
    return x.new_empty(x.shape)


def _loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.mark_non_differentiable(output[1])
    ctx.set_materialize_grads(False)


def _loss_backward(ctx, gout, ggrad=None):
    (grad,) = ctx.saved_tensors
    if gout is None:
        return (None,) * 7
    if grad.numel() == 0:
        raise RuntimeError("paradis_loss was called with want_grad=False but its gradient is requested")
    return (_scale(grad, gout),) + (None,) * 6


_autograd("paradis_loss", _loss_setup, _loss_backward)


def paradis_loss(pred, target, feature_weights, lat_weights=None, kind="reversed_huber", delta=1.0):
    """mean(feature_weights[c] * lat_weights[h] * l(pred - target)); forward and d/dpred in one pass."""
    code = {"mse": 0, "reversed_huber": 1}[kind]
    require_hip(pred, target, feature_weights, lat_weights)
    want_grad = torch.is_grad_enabled() and pred.requires_grad
    return _paradis_loss(pred, target, feature_weights, lat_weights, code, float(delta), want_grad)[0]


@_define("concat_channels(Tensor[] parts) -> Tensor", autocast=False)
def _concat_channels(parts):
    """cat(parts, dim=1) for [B,C_i,H,W] tensors (channel slices accepted) by strided block copies."""
    _f32(*parts)
    B = parts[0].shape[0]
    H, W = parts[0].shape[-2:]
    chans = [p.shape[1] for p in parts]
    P, Ct = H * W, sum(chans)
    out = torch.empty(B, Ct, H, W, dtype=parts[0].dtype, device=parts[0].device)
    off = 0
    for p, c in zip(parts, chans):
        p, bs = _plane_view(p)
        check(lib.paradis_copy_channels(dptr(p), bs, dptr(out[:, off:]), Ct * P, B, c * P, stream_ptr()),
              "copy_channels")
        off += c
    return out


@_fake("concat_channels")
def _(parts):
    B, _, H, W = parts[0].shape
    return parts[0].new_empty(B, sum(p.shape[1] for p in parts), H, W)


@_define("slice_channels(Tensor x, int off, int c) -> Tensor")
def _slice_channels(x, off, c):
    """contiguous copy of x[:, off:off+c]"""
    _f32(x)
    x = x.contiguous()
    B, Ct, H, W = x.shape
    P = H * W
    g = torch.empty(B, c, H, W, dtype=x.dtype, device=x.device)
    check(lib.paradis_copy_channels(dptr(x[:, off:]), Ct * P, dptr(g), c * P, B, c * P, stream_ptr()),
          "copy_channels")
    return g


@_fake("slice_channels")
def _(x, off, c):
    return x.new_empty(x.shape[0], c, x.shape[2], x.shape[3])


def _cat_setup(ctx, inputs, output):
    ctx.chans = [p.shape[1] for p in inputs[0]]


def _cat_backward(ctx, gout):
    grads, off = [], 0
    need = ctx.needs_input_grad[0]
    for i, c in enumerate(ctx.chans):
        grads.append(_slice_channels(gout, off, c) if need[i] else None)
        off += c
    return (grads,)


_autograd("concat_channels", _cat_setup, _cat_backward)


def concat_channels(parts: Sequence[Tensor]):
    require_hip(*parts)
    return _concat_channels(list(parts))
