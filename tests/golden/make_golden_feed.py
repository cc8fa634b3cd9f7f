#!/usr/bin/env python3
"""Generates tests/golden/f4_feed.pt by importing the REFERENCE's own data-feed functions
(data/forcings/*.py, utils/normalization.py under /root/reference).  Run in the build container only.
Stored: inputs (times as int64 microseconds, lat/lon degrees, raw feature samples) and the reference's
outputs.  numpy version is recorded: the float64-scalar promotion of the TOA accumulation is NEP-50."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from data.forcings import time_forcings, toa_radiation   # noqa: E402
from utils import normalization as N                      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def grid(nlat, nlon, poles):
    if poles:
        lat = np.linspace(-90.0, 90.0, nlat)
    else:
        d = 180.0 / nlat
        lat = -90.0 + d / 2 + d * np.arange(nlat)
    return lat, np.arange(nlon) * (360.0 / nlon)


def main():
    out = {"numpy": np.__version__, "cases": []}
    cases = [("1979-01-01T00", 6, 5, 32, 64, False), ("2020-02-28T18", 6, 4, 32, 64, False),
             ("2019-06-21T03", 1, 7, 33, 64, True), ("2021-12-31T12", 6, 4, 16, 32, False),
             ("2016-09-22T21", 3, 3, 45, 90, True)]
    for start, dt_h, T, nlat, nlon, poles in cases:
        times = np.datetime64(start, "h") + np.arange(T) * np.timedelta64(dt_h, "h")
        for lat_dtype in (np.float64, np.float32):
            lat, lon = grid(nlat, nlon, poles)
            lat, lon = lat.astype(lat_dtype), lon.astype(lat_dtype)
            tf = time_forcings(times)
            rad = toa_radiation(times, lat, lon)
            out["cases"].append({
                "times_us": torch.from_numpy(times.astype("datetime64[us]").astype(np.int64)),
                "lat_deg": torch.from_numpy(lat.copy()), "lon_deg": torch.from_numpy(lon.copy()),
                "time_forcings": {k: torch.from_numpy(np.asarray(v, dtype=np.float64)) for k, v in tf.items()},
                "toa_radiation": torch.from_numpy(rad),
            })
    g = torch.Generator().manual_seed(7)
    x = torch.rand(3, 5, 7, 6, generator=g) * torch.tensor([300.0, 0.02, 0.01, 50.0, 1e-3, 2.0])
    q_min, q_max = torch.tensor(1e-7), torch.tensor(0.025)
    mean, std = torch.tensor(270.0), torch.tensor(15.0)
    out["norm"] = {
        "x": x, "q_min": q_min, "q_max": q_max, "mean": mean, "std": std,
        "standard": N.normalize_standard(x[..., 0], mean, std),
        "humidity": N.normalize_humidity(x[..., 1], q_min, q_max, 1e-12),
        "precip": N.normalize_precipitation(x[..., 2]),
        "de_standard": N.denormalize_standard(N.normalize_standard(x[..., 0], mean, std), mean, std),
        "de_humidity": N.denormalize_humidity(N.normalize_humidity(x[..., 1], q_min, q_max, 1e-12), q_min, q_max, 1e-12),
        "de_precip": N.denormalize_precipitation(N.normalize_precipitation(x[..., 2])),
    }
    torch.save(out, os.path.join(HERE, "f4_feed.pt"))
    print("wrote f4_feed.pt", os.path.getsize(os.path.join(HERE, "f4_feed.pt")) // 1024, "KiB")


if __name__ == "__main__":
    main()
