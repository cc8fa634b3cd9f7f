#!/usr/bin/env python3
"""Measurement of row f4: the forcings of one training batch on the device vs the reference's
numpy arithmetic (oracle/feed_oracle.py, the CPU restatement pinned to the reference) on the host.
Usage on the GPU box: python tools/feed_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import feed_oracle as FO  # noqa: E402
from paradis_model_amd import feed    # noqa: E402
from paradis_model_amd.harness import GRID_PRESETS, make_grids  # noqa: E402


def main():
    for name, S, B in (("5.625deg", 1, 32), ("5.625deg", 6, 32), ("1.40625deg", 1, 8), ("0.25deg", 1, 1)):
        nlat, nlon, poles = GRID_PRESETS[name]
        lat_deg, _, _ = make_grids(nlat, nlon, poles)
        lon_deg = np.arange(nlon) * (360.0 / nlon)
        lat = lat_deg.double().numpy()
        T = S + 1                                      # n_time_inputs = 2
        starts = np.datetime64("2019-01-01T00", "h") + np.arange(B) * np.timedelta64(6 * 7, "h")
        series = [s + np.arange(T) * np.timedelta64(6, "h") for s in starts]
        # device: the whole batch in one launch, grids already resident
        stack = np.stack(series)
        lat_d, lon_d = torch.from_numpy(lat).cuda(), torch.from_numpy(lon_deg).cuda()
        for _ in range(3):
            outs = [feed.compute_forcings(stack, lat_d, lon_d, 2, 250.0, 300.0)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            outs = [feed.compute_forcings(stack, lat_d, lon_d, 2, 250.0, 300.0)]
        torch.cuda.synchronize()
        gpu = (time.perf_counter() - t0) / 10
        ncpu = min(B, 4)
        t0 = time.perf_counter()
        for t in series[:ncpu]:
            FO.compute_forcings(t, lat, lon_deg, 2, 250.0, 300.0)
        cpu = (time.perf_counter() - t0) * B / ncpu
        by = sum(o.numel() for o in outs) * 4
        print(f"{name:10s} S={S} B={B}: device {gpu * 1e3:8.3f} ms/batch ({by / gpu / 1e9:7.1f} GB/s written), "
              f"numpy 1 core {cpu * 1e3:9.1f} ms/batch  ({cpu / gpu:6.0f}x)")


if __name__ == "__main__":
    main()
