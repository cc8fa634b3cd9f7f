"""Data feed on the device: the forcings the reference computes per
sample with numpy in its dataloader workers (``data/era5_dataset.py:587-621`` calling
``data/forcings/time_vars.py`` and ``data/forcings/toa_radiation.py``) and the feature normalisations
of ``utils/normalization.py`` as ``data/era5_dataset.py:547-584`` applies them, as HIP kernels.

No CPU fallback: the tensors must live on the HIP device.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np
import torch

from ._lib import check, dptr, lib, require_hip, stream_ptr

FORCING_CODES = {"toa_incident_solar_radiation": 0, "sin_time_of_day": 1, "cos_time_of_day": 2,
                 "sin_year_progress": 3, "cos_year_progress": 4}
DEFAULT_FORCINGS = tuple(FORCING_CODES)          # order of config/paradis_settings.yaml:214-219
KIND_NONE, KIND_ZSCORE, KIND_HUMIDITY, KIND_PRECIP = 0, 1, 2, 3


def times_to_us(times) -> torch.Tensor:
    """numpy datetime64 array (any unit) -> int64 microseconds since 1970 (CPU tensor)."""
    if isinstance(times, torch.Tensor):
        return times.to(torch.int64)
    return torch.from_numpy(np.asarray(times).astype("datetime64[us]").astype(np.int64))


def _grid64(g, device):
    t = g if isinstance(g, torch.Tensor) else torch.as_tensor(np.asarray(g))
    return t.to(device=device, dtype=torch.float64).contiguous(), int(t.dtype == torch.float32)


def compute_forcings(times, lat_deg, lon_deg, n_time_inputs: int, toa_mean: float, toa_std: float,
                     forcing_inputs: Sequence[str] = DEFAULT_FORCINGS, device="cuda") -> torch.Tensor:
    """``Era5Dataset._compute_forcings`` (reference data/era5_dataset.py:587-621) on the device.

    times: T = steps + n_time_inputs - 1 consecutive timestamps (datetime64 array or int64 us tensor),
    or a [B, T] stack of such series - one launch for a whole batch, each row treated as the dataset
    treats one sample; lat_deg [H], lon_deg [W]: 1-D grids in degrees (numpy or torch, ideally already
    on the device; a float32 latitude array keeps numpy's float32 arithmetic).
    Returns [steps, H, W, len(forcing_inputs) * n_time_inputs] float32 (with a leading B for stacked
    series); names the reference does not know are skipped, as it does."""
    codes = [FORCING_CODES[v] for v in forcing_inputs if v in FORCING_CODES]
    if not codes:
        return None
    t_us = times_to_us(times).to(device).contiguous()
    batched = t_us.dim() == 2
    if t_us.dim() not in (1, 2):
        raise ValueError("times must be [T] or [B, T]")
    B, T = (t_us.shape if batched else (1, t_us.numel()))
    lat64, lat_is_f32 = _grid64(lat_deg, t_us.device)
    lon64, _ = _grid64(lon_deg, t_us.device)
    require_hip(t_us, lat64, lon64, any_dtype=True)
    H, W = lat64.numel(), lon64.numel()
    steps = T - n_time_inputs + 1
    if steps < 1:
        raise ValueError(f"need at least n_time_inputs={n_time_inputs} timestamps, got {T}")
    out = torch.empty(B, steps, H, W, len(codes) * n_time_inputs, dtype=torch.float32, device=t_us.device)
    ws = torch.empty(lib.paradis_forcings_ws_bytes(B, T), dtype=torch.uint8, device=t_us.device)
    arr = (ctypes.c_int * len(codes))(*codes)
    check(lib.paradis_forcings(dptr(t_us), dptr(lat64), dptr(lon64), lat_is_f32, B, T, H, W, n_time_inputs, arr,
                               len(codes), float(toa_mean), float(toa_std), dptr(out), dptr(ws), stream_ptr()),
          "forcings")
    return out if batched else out[0]


def normalize_features_(data: torch.Tensor, kind, p0, p1, eps_q: float = 1e-12, inverse: bool = False):
    """In-place channels-last (de)normalisation of ``data[..., C]`` (reference
    data/era5_dataset.py:547-584 + utils/normalization.py): per channel ``kind`` selects none / z-score
    (p0 mean, p1 std) / specific humidity (p0 q_min, p1 q_max) / precipitation."""
    require_hip(data)
    if not data.is_contiguous() or data.dtype != torch.float32:
        raise ValueError("normalize_features_: contiguous float32 tensor required")
    C = data.shape[-1]
    dev = data.device
    kind_t = torch.as_tensor(kind, dtype=torch.int32, device=dev).contiguous()
    p0_t = torch.as_tensor(p0, dtype=torch.float32, device=dev).contiguous()
    p1_t = torch.as_tensor(p1, dtype=torch.float32, device=dev).contiguous()
    if kind_t.numel() != C or p0_t.numel() != C or p1_t.numel() != C:
        raise ValueError("normalize_features_: kind/p0/p1 need one entry per channel")
    check(lib.paradis_normalize_features(dptr(data), dptr(kind_t), dptr(p0_t), dptr(p1_t), data.numel() // C, C,
                                         float(eps_q), int(inverse), stream_ptr()), "normalize_features")
    return data
