"""ctypes binding of the C-ABI library ``libparadis_hip.so`` (include/paradis_hip.h).

There is NO fallback: if the library is missing the import fails loudly, and every
op requires CUDA(HIP) fp32 tensors.  PyTorch is only used for device memory and
the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PARADIS_HIP_LIB: diagnostic override (A/B of two builds of the same ABI on one box, tools/ab_step.sh)
LIB_PATH = os.environ.get("PARADIS_HIP_LIB") or os.path.join(_HERE, "libparadis_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension is not built. Run "
        "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C paradis_model_amd/csrc`). "
        "paradis_model_amd has no CPU/eager fallback by design.")

lib = ctypes.CDLL(LIB_PATH)

P, I, L, F, S = c_void_p, c_int, c_int64, c_float, c_size_t

# name -> (restype, argtypes); mirrors include/paradis_hip.h one to one
SIGNATURES = {
    "paradis_abi_version": (I, []),
    "paradis_last_error": (c_char_p, []),
    "paradis_geocyclic_pad_fwd": (I, [P, P, L, I, I, I, P]),
    "paradis_geocyclic_pad_bwd": (I, [P, P, L, I, I, I, P]),
    "paradis_sl_advect_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, L, L, L, F, F, F, F, F, I, I, P, P]),
    "paradis_sl_advect_ws_bytes": (S, [I, I, I, I, I]),
    "paradis_sl_advect_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, L, L, L, L, L,
                                  F, F, F, F, F, I, I, P, P]),
    "paradis_dwconv_geo_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "paradis_dwconv_geo_dgrad": (I, [P, P, P, I, I, I, I, I, P]),
    "paradis_dwconv_geo_dgrad_add": (I, [P, P, P, P, I, I, I, I, I, P]),
    "paradis_dwconv_geo_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, P]),
    "paradis_dwconv_geo_wgrad_ws_bytes": (S, [I, I, I, I, I]),
    "paradis_dwconv_geo_wgrad": (I, [P, P, P, P, I, I, I, I, I, P, P]),
    "paradis_avgpool_geo_fwd": (I, [P, P, L, I, I, I, P]),
    "paradis_avgpool_geo_bwd": (I, [P, P, L, I, I, I, P]),
    "paradis_upsample_lonp_fwd": (I, [P, P, L, I, I, I, I, P]),
    "paradis_upsample_lonp_bwd": (I, [P, P, L, I, I, I, I, P]),
    "paradis_amax_partials": (I, [P, I, L, L, P, P]),
    "paradis_pw_gemm_split_bytes": (S, [I, I, I]),
    "paradis_pw_gemm_split_weights": (I, [P, I, I, I, I, P, P]),
    "paradis_pw_gemm_split_weights_pair": (I, [P, I, I, P, P, P]),
    "paradis_pw_gemm_split_weights_pair_scheme": (I, [P, I, I, I, P, P, P]),
    "paradis_pw_gemm_fwd": (I, [P, P, P, I, P, P, P, P, P, P, I, P, P, P, I, I, I, I, L, L, L, I, P]),
    "paradis_pw_gemm_fwd_gated": (I, [P, P, P, I, P, P, P, P, P, P, I, P, P, P, P, I, I, I, I, L, L, L, I, P]),
    "paradis_forcings_ws_bytes": (ctypes.c_size_t, [I, I]),
    "paradis_forcings": (I, [P, P, P, I, I, I, I, I, I, P, I, ctypes.c_double, ctypes.c_double, P, P, P]),
    "paradis_normalize_features": (I, [P, P, P, P, L, I, ctypes.c_float, I, P]),
    "paradis_global_bias_m8_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "paradis_global_bias_m8_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, P, P]),
    "paradis_global_bias_proj_bwd": (I, [P, P, P, P, P, I, I, L, P]),
    "paradis_transpose": (I, [P, P, I, I, P]),
    "paradis_pw_gemm_dgrad": (I, [P, P, I, P, P, P, P, P, I, I, I, I, L, L, L, L, I, P]),
    "paradis_pw_gemm_wgrad_ws_bytes": (S, [I, I, I, I]),
    "paradis_pw_gemm_wgrad_slabs": (I, [I, I, I, I]),
    "paradis_pw_gemm_wgrad": (I, [P, P, P, P, I, I, I, I, L, L, I, P, P, P, P]),
    "paradis_pw_gemm_fwd16": (I, [P, P, P, P, P, P, I, P, P, P, P, I, I, I, I, L, L, L, I, I, P]),
    "paradis_pw_gemm_dgrad16": (I, [P, P, P, P, I, I, I, I, L, L, L, I, I, P]),
    "paradis_pw_gemm_wgrad16": (I, [P, P, P, P, I, I, I, I, L, L, I, P, P]),
    "paradis_bias_grads16": (I, [P, P, P, I, I, I, L, P]),
    "paradis_channel_norm_fwd16": (I, [P, P, P, P, P, P, P, I, I, I, I, L, L, F, P]),
    "paradis_dwconv_geo_fwd16": (I, [P, P, P, P, I, I, I, I, I, P]),
    "paradis_channel_norm_bwd16_ok": (I, [I, I, I]),
    "paradis_channel_norm_bwd16": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, L, L, L, L, P, L, P, P]),
    "paradis_dwconv_geo_bwd16_ok": (I, [I, I, I]),
    "paradis_dwconv_geo_bwd16": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, P]),
    "paradis_channel_norm_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, L, L, F, P]),
    "paradis_channel_norm_bwd_ws_bytes": (S, [I, I, I]),
    "paradis_channel_norm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, L, L, L, L, P, L, P, P]),
    "paradis_global_bias_map_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "paradis_global_bias_map_bwd_ws_bytes": (S, [I, I, I, I, I]),
    "paradis_global_bias_map_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, P]),
    "paradis_act_fwd": (I, [P, P, L, I, P]),
    "paradis_act_bwd": (I, [P, P, P, L, I, P]),
    "paradis_act_bwd16": (I, [P, P, P, L, I, P]),
    "paradis_gated_blend_fwd": (I, [P, P, P, P, I, I, I, P]),
    "paradis_gated_blend_bwd_ws_bytes": (S, [I, I, I]),
    "paradis_gated_blend_bwd": (I, [P, P, P, P, P, P, P, I, I, I, P, P]),
    "paradis_gated_blend_bwd_out": (I, [P, P, P, P, P, P, P, I, I, I, P, P]),
    "paradis_bias_grads": (I, [P, P, P, I, I, I, L, P]),
    "paradis_add": (I, [P, P, P, L, P]),
    "paradis_add_bcast": (I, [P, P, P, L, I, P]),
    "paradis_loss_blocks": (I, [L]),
    "paradis_loss_fwd_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, F, P]),
    "paradis_scale": (I, [P, P, P, L, P]),
    "paradis_copy_channels": (I, [P, L, P, L, I, L, P]),
    "paradis_adamw_step": (I, [P, P, P, P, L, F, F, F, F, F, I, P]),
    "paradis_muon_ws_bytes": (S, [I, I, I]),
    "paradis_muon_step": (I, [P, I, I, I, I, F, F, F, F, F, F, I, I, I, P, P]),
    "paradis_bgemm": (I, [P, P, P, P, I, I, I, I, L, L, L, L, P, P]),
    "paradis_adamw_chunk": (I, []),
    "paradis_adamw_multi": (I, [P, P, P, P, I, I, F, F, F, F, F, I, P, P]),
    "paradis_adamw_tick": (I, [P, P]),
}

_missing = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        _missing.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args
if _missing and os.environ.get("PARADIS_DEV_PARTIAL") != "1":
    raise ImportError(f"{LIB_PATH} lacks symbols {_missing}; rebuild the HIP extension")


def last_error() -> str:
    msg = lib.paradis_last_error()
    return msg.decode() if msg else ""


def check(rc: int, name: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {last_error()}")


def dptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


# the current stream's handle without building a torch.cuda.Stream object per call (7 us each, ~600 calls per training
# step); the private entry point is resolved once, the public one is the fallback
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_hip(*tensors, any_dtype: bool = False) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("paradis_model_amd ops run on the MI355X only: got a CPU tensor "
                               "(there is no CPU fallback; use oracle/ for CPU checks)")
        if t.dtype != torch.float32 and not any_dtype:
            raise RuntimeError(f"paradis_model_amd ops are fp32; got {t.dtype}")


class LaunchProfiler:
    """Optional HIP-event timing of selected C-ABI calls on the current stream (used by bench.py
    for the roofline lines).  ``work`` is the algorithmic FLOP or byte count of the call."""

    def __init__(self):
        self.records = {}     # name -> [work_sum, [(ev0, ev1), ...]]

    def add(self, name, work, ev0, ev1):
        rec = self.records.setdefault(name, [0.0, []])
        rec[0] += work
        rec[1].append((ev0, ev1))

    def summary(self):
        """name -> dict(launches, work, ms) ; call after torch.cuda.synchronize()"""
        out = {}
        for name, (work, evs) in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            out[name] = {"launches": len(evs), "work": work, "ms": ms}
        return out


PROFILER = None


def call(name: str, work: float, *args, symbol: str = None) -> None:
    """Invoke ``paradis_<name>`` (or ``paradis_<symbol>``, accounted under ``name``) and raise on a non-zero return
    code; time it when profiling."""
    fn = getattr(lib, "paradis_" + (symbol or name))
    prof = PROFILER
    if prof is None:
        check(fn(*args), name)
        return
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    rc = fn(*args)
    ev1.record()
    check(rc, name)
    prof.add(name, work, ev0, ev1)
