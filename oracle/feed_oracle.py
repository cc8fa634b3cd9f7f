"""CPU restatement of the reference's data-feed arithmetic: temporal and
top-of-atmosphere-radiation forcings and the feature normalisations.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product path).  Pinned against
outputs of the reference's own functions (tests/golden/make_golden_feed.py imports
``data.forcings`` and ``utils.normalization`` from /root/reference and stores inputs/outputs in
tests/golden/f4_feed.pt).

Every function cites the reference lines it follows.  numpy promotion semantics are those of
numpy >= 2 (NEP 50), the version the goldens were generated with: a ``numpy.float64`` *scalar*
times a float32 array gives float64.
"""
from __future__ import annotations

import numpy as np
import torch

JULIAN_REF_US = np.datetime64("2000-01-01T12:00", "us").astype(np.float64)   # toa_radiation.py:30-31
QNODES, QWEIGHTS = np.polynomial.legendre.leggauss(15)                        # toa_radiation.py:163
FORCING_ORDER = ("toa_incident_solar_radiation", "sin_time_of_day", "cos_time_of_day",
                 "sin_year_progress", "cos_year_progress")                   # config/paradis_settings.yaml:214-219


def time_forcings(times: np.ndarray) -> dict:
    """data/forcings/time_vars.py:6-40 (sin/cos of hour-of-day/24 and day-of-year/365.25, float64)."""
    t_h = times.astype("datetime64[h]")
    hour = (t_h - t_h.astype("datetime64[D]")) / np.timedelta64(1, "h")
    tod = hour / 24
    doy = (t_h - t_h.astype("datetime64[Y]")) / np.timedelta64(1, "D")
    yp = doy / 365.25
    return {"sin_time_of_day": np.sin(2 * np.pi * tod), "cos_time_of_day": np.cos(2 * np.pi * tod),
            "sin_year_progress": np.sin(2 * np.pi * yp), "cos_year_progress": np.cos(2 * np.pi * yp)}


def solar_scalars(time_us: float):
    """Per-instant scalars of toa_radiation.py:38-76,139-152: (declination f32, mod_day f32,
    irradiance 1360.56/d^2 f64) in float64 arithmetic."""
    mjd = (time_us - JULIAN_REF_US) / 86400e6
    anomaly = np.mod(357.529 + 0.98560028 * mjd, 360) * np.pi / 180
    mean_lon = np.mod(280.459 + 0.98564736 * mjd, 360) * np.pi / 180
    app_lon = mean_lon + (1.915 * np.sin(anomaly) + 0.020 * np.sin(2 * anomaly)) * np.pi / 180
    dist = 1.00014 - 0.01671 * np.cos(anomaly) - 0.00014 * np.cos(2 * anomaly)
    obliq = (23.439 - 0.00000036 * mjd) * np.pi / 180
    asc = np.arctan2(np.cos(obliq) * np.sin(app_lon), np.cos(app_lon))
    decl = np.arcsin(np.sin(obliq) * np.sin(app_lon))
    eot = (np.mod(mean_lon - asc + np.pi, 2 * np.pi) - np.pi) / (2 * np.pi)      # :80-83,146-148
    mod_day = np.float32(np.mod(mjd + eot, 1) * 2 * np.pi)                       # :93
    return np.float32(decl), mod_day, 1360.56 / dist ** 2


def toa_radiation(times: np.ndarray, lat_deg: np.ndarray, lon_deg: np.ndarray) -> np.ndarray:
    """toa_radiation.py:172-199 -> :166-169 -> :126-159: radiation integrated over the hour ending at
    each time by 15-point Gauss-Legendre, float32 [T,H,W]."""
    lat_rad = (np.asarray(lat_deg).reshape(-1, 1) * np.pi / 180).astype(np.float32)   # :135
    lon32 = np.asarray(lon_deg).reshape(1, -1).astype(np.float32)                    # :136
    slat, clat = np.sin(lat_rad[:, 0]), np.cos(lat_rad[:, 0])
    out = np.empty((len(times), lat_rad.shape[0], lon32.shape[1]), dtype=np.float32)
    for k, t in enumerate(np.asarray(times).astype("datetime64[us]").astype(np.float64)):
        acc = np.zeros(out.shape[1:], dtype=np.float32)
        for q, w in zip(t - 3600e6 * (1 + QNODES) / 2, 3600 * QWEIGHTS / 2):          # :167-168
            decl, mod_day, irr = solar_scalars(q)
            lst = (lon32[0] * np.pi / 180 + mod_day).astype(np.float32)              # :88-95 (float32)
            cz = np.maximum(0, slat[:, None] * np.sin(decl) + clat[:, None] * np.cos(decl) * np.cos(lst))
            acc += cz * (irr * w)          # :117-123,153-158: float64 scalar weight, float32 accumulate
        out[k] = acc
    return out


def compute_forcings(times, lat_deg, lon_deg, n_time_inputs: int, toa_mean: float, toa_std: float,
                     forcing_inputs=FORCING_ORDER) -> torch.Tensor:
    """data/era5_dataset.py:587-621: [steps,H,W,len(forcings)*n_time_inputs] float32, steps =
    len(times) - n_time_inputs + 1; per variable the last axis holds the n_time_inputs window."""
    tf = time_forcings(np.asarray(times))
    H, W = len(lat_deg), len(lon_deg)
    steps = len(times) - n_time_inputs + 1
    parts = []
    for var in forcing_inputs:
        if var == "toa_incident_solar_radiation":
            rad = toa_radiation(times, lat_deg, lon_deg)
            t = torch.tensor((rad - toa_mean) / toa_std, dtype=torch.float32)
            parts.append(t.unfold(0, n_time_inputs, 1))
        elif var in tf:
            v = torch.tensor(tf[var], dtype=torch.float32).unfold(0, n_time_inputs, 1)
            parts.append(v.view(steps, 1, 1, n_time_inputs).expand(steps, H, W, n_time_inputs))
    return torch.cat(parts, dim=-1)


# ---- utils/normalization.py -------------------------------------------------------------------
def normalize_standard(x, mean, std):            # :6-8
    return (x - mean) / std


def denormalize_standard(x, mean, std):          # :11-13
    return x * std + mean


def normalize_humidity(x, q_min, q_max, eps=1e-12):   # :16-37
    return (torch.log(torch.clip(x, 0, q_max) + eps) - torch.log(q_min)) / (torch.log(q_max) - torch.log(q_min))


def denormalize_humidity(x, q_min, q_max, eps=1e-12):  # :40-53
    q = torch.exp(x * (torch.log(q_max) - torch.log(q_min)) + torch.log(q_min)) - eps
    return torch.clip(q, min=0, max=q_max)


def normalize_precipitation(x, shift=10, eps=1e-6):    # :56-67
    return torch.log(x + eps) + shift


def denormalize_precipitation(x, shift=10, eps=1e-6):  # :70-80
    return torch.clip(torch.exp(x - shift) - eps, min=0)


KIND_NONE, KIND_ZSCORE, KIND_HUMIDITY, KIND_PRECIP = 0, 1, 2, 3


def normalize_features(x: torch.Tensor, kind, p0, p1, eps_q=1e-12, inverse=False) -> torch.Tensor:
    """Channels-last feature normalisation as data/era5_dataset.py:547-584 applies it: per channel c,
    kind[c] selects z-score (p0=mean, p1=std), humidity (p0=q_min, p1=q_max) or precipitation."""
    out = x.clone()
    for c in range(x.shape[-1]):
        k = int(kind[c])
        a, b = torch.as_tensor(p0[c], dtype=x.dtype), torch.as_tensor(p1[c], dtype=x.dtype)
        if k == KIND_ZSCORE:
            out[..., c] = denormalize_standard(x[..., c], a, b) if inverse else normalize_standard(x[..., c], a, b)
        elif k == KIND_HUMIDITY:
            out[..., c] = (denormalize_humidity(x[..., c], a, b, eps_q) if inverse
                           else normalize_humidity(x[..., c], a, b, eps_q))
        elif k == KIND_PRECIP:
            out[..., c] = denormalize_precipitation(x[..., c]) if inverse else normalize_precipitation(x[..., c])
    return out
