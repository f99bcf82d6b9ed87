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


@pytest.mark.parametrize("H,W,P", [(720, 1440, 4), (1800, 3600, 10), (450, 900, 3), (16, 32, 4)])
def test_handle_lat_lon_tables_vs_torch_derived_tables(H, W, P):
    """A non-Python caller of the C ABI passes lat / lon and lets `aurora_hip_precompute` derive the position / scale
    tables itself (C++: float32 geometry, fp64 trigonometry) instead of handing in torch-computed ones.  Stated bound
    per feature column, with x the encoded quantity and lambda the column's wavelength: the two derivations may differ
    by a few float32 ulps of x (torch's vectorised float32 sin / sqrt / pooling vs libm's), i.e. by
    2 pi ulp(x) / lambda radians of phase -- negligible for long wavelengths, O(1) for the shortest scale wavelengths
    (1e-4 km against root areas of ~28 km): those columns are chaotic in the reference itself.
    Host-only code: runs without a GPU."""
    import ctypes

    from aurora_amd.engine import lib

    D = 512
    lat = np.linspace(90, -90, H + 1)[:-1].astype(np.float32).astype(np.float64)
    lon = np.linspace(0, 360, W + 1)[:-1].astype(np.float32).astype(np.float64)
    L = (H // P) * (W // P)
    pos, scale = np.empty((L, D), np.float32), np.empty((L, D), np.float32)
    code = lib.load().aurora_hip_pos_scale_encoding(lat.ctypes.data_as(lib._PD), lon.ctypes.data_as(lib._PD), H, W, P, D,
                                                    pos.ctypes.data_as(lib._PF), scale.ctypes.data_as(lib._PF))
    assert code == 0, lib.load().aurora_hip_last_error()
    pos_t, scale_t = encodings.pos_scale_encodings(D, lat.astype(np.float32), lon.astype(np.float32), P)

    def bound(kind, x_max, d):
        lower, upper, _ = encodings._EXPANSIONS[kind]
        lam = np.power(10.0, np.linspace(np.log10(lower), np.log10(upper), d // 2))
        ulps = 4 * np.spacing(np.float32(x_max)).astype(np.float64)
        return np.minimum(2.0, 2 * np.pi * ulps / lam + 2e-7)   # phase error (rad) bounds |sin / cos difference|

    # position: patch-mean latitude (<= 90) | longitude (< 360), D/4 wavelengths each, sin | cos halves
    q = D // 4
    for lo, x_max in ((0, 90.0), (D // 2, 360.0)):
        b = bound("pos", x_max, D // 2)
        for half in (0, q):
            diff = np.abs(pos[:, lo + half:lo + half + q] - pos_t[:, lo + half:lo + half + q]).max(axis=0)
            assert (diff <= b).all(), (lo, half, np.argmax(diff - b), diff.max())
    # scale: x = root area of a patch = sqrt(R^2 pi (sin(lat_max) - sin(lat_min)) dlon) evaluated in float32: the
    # difference of two float32 sines cancels, so one ulp of a sine (2^-24) is a relative error of 2^-24 / dsin in the
    # area -- up to 4e-4 next to the poles.  Bound per patch l and wavelength j: 2 pi dx_l / lambda_j with
    # dx_l = x_l * (2^-23 / dsin_l + 2^-21) / 2.
    Hp, Wp = H // P, W // P
    lat_r = np.deg2rad(lat[:Hp * P].reshape(Hp, P))
    dsin = np.sin(lat_r.max(axis=1)) - np.sin(lat_r.min(axis=1))
    x_ref = np.sqrt(6371.0 ** 2 * np.pi * dsin[:, None] * np.deg2rad(lon[P - 1] - lon[0]) * np.ones((1, Wp))).reshape(-1)
    dx = x_ref * (2.0 ** -23 / np.repeat(dsin, Wp) + 2.0 ** -21) / 2
    lower, upper, _ = encodings._EXPANSIONS["scale"]
    lam = np.power(10.0, np.linspace(np.log10(lower), np.log10(upper), D // 2))
    b = np.minimum(2.0, 2 * np.pi * dx[:, None] / lam[None, :] + 2e-7)
    for half in (0, D // 2):
        diff = np.abs(scale[:, half:half + D // 2] - scale_t[:, half:half + D // 2])
        assert (diff <= b).all(), (half, np.unravel_index(np.argmax(diff - b), diff.shape), diff.max())
    # and the long-wavelength columns, where the bound is tight, really are equal to float32 rounding
    smooth = b.max(axis=0) < 1e-5
    assert smooth.sum() >= D // 8
    assert np.abs(scale[:, :D // 2][:, smooth] - scale_t[:, :D // 2][:, smooth]).max() < 1e-5
