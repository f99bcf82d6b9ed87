"""aurora_amd.Tracker against a track the REFERENCE tracker produced (tools/make_tracker_golden.py runs aurora/tracker.py
on the synthetic storm of tests/tracker_scenario.py): every fix, minimum pressure and maximum wind must be EQUAL -- the
window arithmetic is the reference's on the same float32 values.  The scenario crosses land (geopotential fall-back),
loses the eye (extrapolation + failure count) and, in its second track, the 360 -> 0 seam (wrapped windows)."""
import json
from datetime import datetime
from pathlib import Path

import numpy as np
import pytest
import torch

import aurora_amd
from aurora_amd import Batch, Metadata
from aurora_amd.tracker import NoEyeException, extrapolate
from tests import tracker_scenario as sc

GOLD = json.loads((Path(__file__).parent / "golden" / "tracker_track.json").read_text())


def run(name, device):
    lon0 = sc.SCENARIOS[name]
    tracker = aurora_amd.Tracker(init_lat=sc.START[0], init_lon=lon0, init_time=sc.START[2])
    for step in range(1, sc.STEPS + 1):
        tracker.step(sc.batch(step, Batch, Metadata, device=device, lon0=lon0))
    return tracker


def check(tracker, name):
    ref, df = GOLD[name], tracker.results()
    assert tracker.fails == ref["fails"] == 1
    assert [t.isoformat() for t in df["time"]] == ref["time"]
    for k in ("lat", "lon", "msl", "wind"):
        want = np.array([np.nan if x is None else x for x in ref[k]])
        np.testing.assert_array_equal(np.asarray(df[k], dtype=np.float64), want, err_msg=k)


@pytest.mark.parametrize("name", list(sc.SCENARIOS))
def test_track_equals_the_reference_tracker(name):
    check(run(name, "cpu"), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(sc.SCENARIOS))
def test_track_of_device_resident_predictions(name):
    """Predictions stay on the device; only the windows are copied."""
    check(run(name, "cuda"), name)


def test_interface_and_errors_as_upstream():
    t = aurora_amd.Tracker(init_lat=10.0, init_lon=20.0, init_time=datetime(2020, 1, 1))
    df = t.results()
    assert list(df.columns) == ["time", "lat", "lon", "msl", "wind"] and len(df) == 1 and np.isnan(df["msl"][0])
    two = sc.batch(1, Batch, Metadata)
    two = Batch(two.surf_vars, two.static_vars, two.atmos_vars,
                Metadata(two.metadata.lat, two.metadata.lon, two.metadata.time * 2, two.metadata.atmos_levels))
    with pytest.raises(RuntimeError, match="batch size one"):
        t.step(two)
    with pytest.raises(ValueError):
        extrapolate([], [])
    assert extrapolate([3.0], [4.0]) == (3.0, 4.0)
    lat, lon = extrapolate([0.0, 1.0, 2.0], [10.0, 12.0, 14.0])
    assert abs(lat - 3.0) < 1e-9 and abs(lon - 16.0) < 1e-9
    # smooth tilted fields at the very first step: nothing to extrapolate from -> the reference's exception
    first = aurora_amd.Tracker(init_lat=sc.START[0] + 8.1, init_lon=sc.START[1] + 14.0, init_time=sc.START[2])
    with pytest.raises(NoEyeException):
        first.step(sc.batch(9, Batch, Metadata))
