"""Shared helpers for tests: seeded inputs (same recipe as tests/golden/make_golden.py),
grids, error metrics."""
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def seeded(seed, *shape, scale=1.0, kind="randn"):
    g = torch.Generator().manual_seed(seed)
    if kind == "randn":
        return torch.randn(*shape, generator=g) * scale
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def chk(t):
    return [float(t.double().sum()), float(t.double().abs().sum())]


def assert_chk(tensors, expected):
    got = []
    for t in tensors:
        got += chk(t)
    for a, b in zip(got, expected):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), "seeded input recipe drifted from golden"


def make_grid(nlat, nlon, poles):
    """lat ascending; same construction as the golden generator / data/era5_dataset.py:178-182"""
    if poles:
        lat = torch.linspace(-90.0, 90.0, nlat, dtype=torch.float64)
    else:
        d = 180.0 / nlat
        lat = -90.0 + d / 2 + d * torch.arange(nlat, dtype=torch.float64)
    lon = torch.arange(nlon, dtype=torch.float64) * (360.0 / nlon)
    lat_r = torch.deg2rad(lat.to(torch.float32))
    lon_r = torch.deg2rad(lon.to(torch.float32))
    lg, og = torch.meshgrid(lat_r, lon_r, indexing="ij")
    return lat.to(torch.float32), lg.contiguous(), og.contiguous()


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


def max_rel(a, b):
    """max-abs(diff) / max-abs(ref): the tolerance metric of SURVEY.md section 8c"""
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def rms_rel(a, b):
    a, b = a.detach(), b.detach()
    d = (a.double() - b.double())
    return float(d.pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-30))
