"""GPU parity of the on-device data feed (row f4) against the reference's outputs (golden f4_feed.pt)
and the CPU oracle.  Tolerances: temporal forcings 1e-6 absolute (float64 sin/cos of ocml vs glibc,
rounded to float32); TOA radiation 1e-5 relative to its maximum (float32 sin/cos of 15 quadrature
points: ocml vs numpy's SIMD loops); normalisations 1e-6 relative (float32 log/exp)."""
import numpy as np
import pytest
import torch

from oracle import feed_oracle as FO
from tests._util import load_golden, max_rel

pytestmark = pytest.mark.gpu


def _times(case):
    return case["times_us"].numpy().astype("datetime64[us]")


@pytest.mark.parametrize("n_time_inputs", [1, 2, 3])
def test_forcings_vs_reference_golden(n_time_inputs):
    from paradis_model_amd import feed
    for case in load_golden("f4_feed.pt")["cases"]:
        times, lat, lon = _times(case), case["lat_deg"].numpy(), case["lon_deg"].numpy()
        if len(times) < n_time_inputs:
            continue
        got = feed.compute_forcings(times, lat, lon, n_time_inputs, 0.0, 1.0).cpu()
        n, steps = n_time_inputs, len(times) - n_time_inputs + 1
        assert got.shape == (steps, len(lat), len(lon), 5 * n)
        rad = case["toa_radiation"]
        for k in range(n):
            assert max_rel(got[..., k], rad[k:k + steps]) <= 1e-5
            for v, name in enumerate(("sin_time_of_day", "cos_time_of_day", "sin_year_progress",
                                      "cos_year_progress"), start=1):
                want = case["time_forcings"][name].float()[k:k + steps]
                assert (got[:, :, :, v * n + k] - want.view(-1, 1, 1)).abs().max() <= 1e-6, name


def test_forcings_vs_oracle_normalised_and_subset():
    from paradis_model_amd import feed
    case = load_golden("f4_feed.pt")["cases"][2]
    times, lat, lon = _times(case), case["lat_deg"].numpy(), case["lon_deg"].numpy()
    want = FO.compute_forcings(times, lat, lon, 2, 251.3, 302.7)
    got = feed.compute_forcings(times, lat, lon, 2, 251.3, 302.7).cpu()
    assert max_rel(got, want) <= 1e-5
    order = ("cos_year_progress", "toa_incident_solar_radiation", "not_a_forcing", "sin_time_of_day")
    want = FO.compute_forcings(times, lat, lon, 2, 0.0, 1.0, order)
    got = feed.compute_forcings(times, lat, lon, 2, 0.0, 1.0, order).cpu()
    assert got.shape == want.shape and max_rel(got, want) <= 1e-5
    # calendar edge cases: leap day, year boundary, pre-1970 timestamps
    t = np.array(["2020-02-29T23", "2020-03-01T00", "1999-12-31T23", "2000-01-01T00", "1969-07-20T20",
                  "1900-03-01T06"], dtype="datetime64[h]")
    want = FO.compute_forcings(t, lat[:4], lon[:6], 1, 0.0, 1.0)
    got = feed.compute_forcings(t, lat[:4], lon[:6], 1, 0.0, 1.0).cpu()
    assert (got[..., 1:] - want[..., 1:]).abs().max() <= 1e-6
    assert max_rel(got[..., 0], want[..., 0]) <= 1e-5
    with pytest.raises(ValueError):
        feed.compute_forcings(t[:1], lat, lon, 2, 0.0, 1.0)
    # a [B,T] stack in one launch == the per-series results (windows do not straddle series)
    stack = np.stack([times[0] + np.arange(4) * np.timedelta64(h, "h") for h in (6, 1, 24)])
    got = feed.compute_forcings(stack, torch.from_numpy(lat).cuda(), torch.from_numpy(lon).cuda(), 2, 10.0, 20.0)
    assert got.shape == (3, 3, len(lat), len(lon), 10)
    for b in range(3):
        one = feed.compute_forcings(stack[b], lat, lon, 2, 10.0, 20.0)
        assert torch.equal(got[b], one)


def test_normalise_features_vs_reference_golden():
    from paradis_model_amd import feed
    n = load_golden("f4_feed.pt")["norm"]
    x = n["x"]
    kind = [feed.KIND_ZSCORE, feed.KIND_HUMIDITY, feed.KIND_PRECIP, feed.KIND_NONE, feed.KIND_NONE, feed.KIND_ZSCORE]
    p0 = [270.0, 1e-7, 0, 0, 0, 1.0]
    p1 = [15.0, 0.025, 0, 0, 0, 0.5]
    y = feed.normalize_features_(x.clone().cuda(), kind, p0, p1).cpu()
    assert max_rel(y[..., 0], n["standard"]) <= 1e-6
    assert max_rel(y[..., 1], n["humidity"]) <= 1e-6
    assert max_rel(y[..., 2], n["precip"]) <= 1e-6
    assert torch.equal(y[..., 3], x[..., 3]) and torch.equal(y[..., 4], x[..., 4])
    back = feed.normalize_features_(y.clone().cuda(), kind, p0, p1, inverse=True).cpu()
    assert max_rel(back[..., 0], n["de_standard"]) <= 1e-6
    assert max_rel(back[..., 1], n["de_humidity"]) <= 2e-5      # exp() amplifies the ulp of the log
    assert max_rel(back[..., 2], n["de_precip"]) <= 2e-5
    with pytest.raises(RuntimeError):
        feed.normalize_features_(x.clone(), kind, p0, p1)          # CPU tensor: no fallback
