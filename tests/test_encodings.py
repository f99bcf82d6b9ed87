"""Host-side Fourier tables equal the oracle's (bitwise for the chaotic high frequencies)."""
from datetime import datetime

import numpy as np
import pytest
import torch

from aurora_amd.engine import encodings
from oracle import aurora_oracle as oracle


def test_constants():
    assert encodings.MIN_PATCH_AREA == pytest.approx(oracle.MIN_PATCH_AREA, rel=1e-12)
    assert encodings.MIN_PATCH_AREA == pytest.approx(0.00010814085263058073, rel=1e-12)
    assert encodings.AREA_EARTH == pytest.approx(511207893.39581096, rel=1e-14)


@pytest.mark.parametrize("kind,x", [
    ("lead_time", [6.0]), ("lead_time", [12.0]), ("levels", [50, 100, 850, 1000]),
    ("levels", [50.0, 0.5]), ("pos", [0.0, 0.125, 359.875, -89.875]),
    ("absolute_time", [datetime(2020, 6, 1, 12).timestamp() / 3600]),
])
def test_fourier_matches_oracle(kind, x):
    xt = torch.tensor(x) if kind != "absolute_time" else torch.tensor(x, dtype=torch.float32)
    ref = oracle.fourier_expansion(kind, xt, 64).numpy()
    arr = np.asarray(x, dtype=np.float32) if xt.dtype == torch.float32 else np.asarray(x)
    mine = encodings.fourier(kind, arr, 64)
    assert mine.dtype == np.float32 and mine.shape == ref.shape
    assert np.abs(mine - ref).max() < 1e-6


def test_fourier_range_assertion():
    with pytest.raises(AssertionError):
        encodings.fourier("lead_time", np.asarray([1e6]), 8)
    with pytest.raises(ValueError):
        encodings.fourier("lead_time", np.asarray([6.0]), 7)
    encodings.fourier("absolute_time", np.asarray([1e9]), 8)  # no range check


@pytest.mark.parametrize("H,W,P", [(16, 32, 4), (720, 1440, 4), (40, 80, 10), (24, 48, 3)])
def test_pos_scale_bitwise(H, W, P):
    lat = torch.linspace(90, -90, H + 1)[:-1]
    lon = torch.linspace(0, 360, W + 1)[:-1]
    pos_ref, scale_ref = oracle.pos_scale_encodings(128, lat, lon, P)
    pos, scale = encodings.pos_scale_encodings(128, lat.numpy(), lon.numpy(), P)
    assert np.abs(pos - pos_ref.numpy()).max() < 1e-6
    assert np.abs(scale - scale_ref.numpy()).max() < 1e-6
    # matrix-valued lat/lon give the same tables (tests/test_model.py:126-160 upstream)
    glat, glon = lat[:, None].expand(-1, W), lon[None, :].expand(H, -1)
    pos2, scale2 = encodings.pos_scale_encodings(128, glat, glon, P)
    assert np.array_equal(pos, pos2) and np.array_equal(scale, scale2)


def test_level_and_time_helpers():
    lv = encodings.levels((100, 250, 500, 850), 32)
    assert lv.shape == (4, 32)
    ref = oracle.fourier_expansion("levels", torch.tensor((100, 250, 500, 850)), 32).numpy()
    assert np.abs(lv - ref).max() < 1e-6
    at = encodings.absolute_time([datetime(2020, 6, 1, 12).timestamp() / 3600], 32)
    ref = oracle.fourier_expansion(
        "absolute_time", torch.tensor([datetime(2020, 6, 1, 12).timestamp() / 3600], dtype=torch.float32), 32
    ).numpy()
    assert np.abs(at - ref).max() < 1e-6
    assert np.abs(encodings.lead_time(6.0, 32) - oracle.fourier_expansion(
        "lead_time", torch.tensor([6.0]), 32).numpy()[0]).max() < 1e-6
