"""Fourier feature tables, computed on the host in fp64 and cached.

The reference recomputes these with torch on the device at every step
(aurora/model/fourier.py:45-92, aurora/model/posencoding.py:61-192); they depend only on the
grid, the pressure levels and the clock, so the engine evaluates them once in numpy (fp64, then
cast to fp32 exactly like `encoding.float()` upstream) and keeps the embedded results on the
device.  tests/test_encodings.py compares every function here with the oracle.
"""

from __future__ import annotations

import math
from typing import Sequence

import numpy as np

RADIUS_EARTH_KM = 6378137 / 1000


def _polygon_area_km2(poly: np.ndarray) -> float:
    """Spherical polygon area as the reference computes it (aurora/area.py:12-48), including
    its way of closing the ring by repeating the LAST vertex."""
    pts = np.concatenate((poly, poly[-1:]), axis=0)
    n = len(pts)
    total = 0.0
    for i in range(n):
        total += (math.radians(pts[(i + 2) % n][1]) - math.radians(pts[i][1])) * math.sin(
            math.radians(pts[(i + 1) % n][0])
        )
    return abs(total * RADIUS_EARTH_KM * RADIUS_EARTH_KM / 2)


_DELTA = 0.01
MIN_PATCH_AREA = _polygon_area_km2(
    np.array([[90, 0], [90, _DELTA], [90 - _DELTA, _DELTA], [90 - _DELTA, 0]], dtype=np.float64)
)
AREA_EARTH = 4 * np.pi * RADIUS_EARTH_KM * RADIUS_EARTH_KM

# name -> (lower wavelength, upper wavelength, assert range)      (fourier.py:112-126)
_EXPANSIONS = {
    "pos": (_DELTA, 720.0, True),
    "scale": (MIN_PATCH_AREA, AREA_EARTH, True),
    "lead_time": (1 / 60, 24 * 7 * 3, True),
    "levels": (0.01, 1e5, True),
    "absolute_time": (1.0, 24 * 365.25, False),
}


def fourier(kind: str, x, d: int) -> np.ndarray:
    """(…, d) float32 features [sin(2 pi x / lambda_j), cos(2 pi x / lambda_j)], lambda log-spaced."""
    lower, upper, check = _EXPANSIONS[kind]
    x = np.asarray(x, dtype=np.float64)
    ax = np.abs(x)
    if check and not np.all(((lower <= ax) & np.all(ax <= upper)) | (x == 0)):
        raise AssertionError(
            f"The input tensor is not within the configured range `[{lower}, {upper}]`."
        )
    if d % 2:
        raise ValueError("The dimensionality must be a multiple of two.")
    # torch.logspace(log10(lower), log10(upper), d/2, base=10) in fp64
    exps = np.linspace(math.log10(lower), math.log10(upper), d // 2, dtype=np.float64)
    wavelengths = np.power(10.0, exps)
    prod = x[..., None] * (2 * np.pi / wavelengths)
    return np.concatenate((np.sin(prod), np.cos(prod)), axis=-1).astype(np.float32)


def pos_scale_encodings(d: int, lat, lon, patch: int) -> tuple[np.ndarray, np.ndarray]:
    """(L, d) position and scale encodings of the patch grid from lat/lon (vectors or matrices),
    following posencoding.py:61-192: patch-mean position and patch root area, both in fp32.

    The fp32 pre-processing (pooling, deg2rad, sin, sqrt) runs through torch's CPU kernels on
    purpose: the expansion multiplies its input by up to 2*pi/1e-4, so a 1-ulp fp32 difference
    in the root area moves the highest-frequency features by O(1) rad.  Using the very same
    fp32 kernels as the reference keeps these tables bit-identical to it (one-time host work,
    O(H*W) scalars, cached per grid).
    """
    import torch
    import torch.nn.functional as F

    lat = torch.as_tensor(lat).detach().to("cpu", torch.float32)
    lon = torch.as_tensor(lon).detach().to("cpu", torch.float32)
    if lat.dim() == 1 and lon.dim() == 1:
        glat = lat[:, None].expand(-1, lon.shape[0]).contiguous()
        glon = lon[None, :].expand(lat.shape[0], -1).contiguous()
    elif lat.dim() == 2 and lon.dim() == 2:
        glat, glon = lat.contiguous(), lon.contiguous()
    else:
        raise ValueError(
            "Latitudes and longitudes must either both be vectors or both be matrices, "
            f"but have dimensionalities {lat.dim()} and {lon.dim()} respectively."
        )
    glat, glon = glat[None, None], glon[None, None]
    k = (patch, patch)
    mid_lat, mid_lon = F.avg_pool2d(glat, k)[0, 0], F.avg_pool2d(glon, k)[0, 0]
    lat_max, lat_min = F.max_pool2d(glat, k)[0, 0], -F.max_pool2d(-glat, k)[0, 0]
    lon_max, lon_min = F.max_pool2d(glon, k)[0, 0], -F.max_pool2d(-glon, k)[0, 0]
    assert bool((lat_max > lat_min).all()) and bool((lon_max > lon_min).all())
    area = (
        6371**2 * torch.pi
        * (torch.sin(torch.deg2rad(lat_max)) - torch.sin(torch.deg2rad(lat_min)))
        * (torch.deg2rad(lon_max) - torch.deg2rad(lon_min))
    )
    assert bool((area > 0).all())
    root_area = torch.sqrt(area)
    assert d % 4 == 0
    pos = np.concatenate(
        (fourier("pos", mid_lat.reshape(-1).numpy(), d // 2),
         fourier("pos", mid_lon.reshape(-1).numpy(), d // 2)), axis=-1
    )
    scale = fourier("scale", root_area.reshape(-1).numpy(), d)
    return pos, scale


def lead_time(hours: float, d: int) -> np.ndarray:
    """(d,) encoding of the lead time in hours."""
    return fourier("lead_time", np.asarray([hours], dtype=np.float32), d)[0]


def levels(levels_hpa: Sequence[float], d: int) -> np.ndarray:
    """(n_levels, d) pressure-level encodings (levels are integer/float hPa values)."""
    lv = list(levels_hpa)
    # torch.tensor(levels): int64 for all-int tuples, float32 as soon as one level is a float.
    arr = np.asarray(lv, dtype=np.int64 if all(isinstance(v, (int, np.integer)) for v in lv) else np.float32)
    return fourier("levels", arr, d)


def absolute_time(stamps_hours: Sequence[float], d: int) -> np.ndarray:
    """(B, d) encodings of absolute times given in hours since the epoch.

    The reference converts the timestamps to a float32 tensor before expanding
    (encoder.py:359-362), which quantises them to ~1/32 h around 2020; same here.
    """
    return fourier("absolute_time", np.asarray(list(stamps_hours), dtype=np.float32), d)
