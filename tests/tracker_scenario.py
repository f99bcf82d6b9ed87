"""A deterministic synthetic storm for the tracker tests: shared by tools/make_tracker_golden.py (which feeds it to the
REFERENCE tracker and stores the track) and tests/test_tracker.py (which feeds it to aurora_amd.Tracker).

0.5-degree global grid; a pressure low with a matching 700 hPa geopotential low and a wind ring moves north-west from
(12 N, 176 E) across the date line, over an island (land-sea mask 1: the pressure search is refused there and the tracker
falls back to geopotential), weakens to nothing for two steps (one with smooth fields: no eye at all, the fix is extrapolated and a failure is counted;
one with noise only: a spurious minimum is taken) and re-forms.  Seeded noise makes
the smoothing and the 8 x 8 minimum filter matter.
"""
from datetime import datetime, timedelta

import numpy as np
import torch

STEPS = 16
START = (12.0, 176.0, datetime(2022, 9, 1, 0))
LEVELS = (500, 700, 850)


SCENARIOS = {"pacific": 176.0, "greenwich": 351.0}   # start longitude: the second track crosses 360 -> 0 (wrapped windows)


def fields(step: int, lon0: float = START[1]):
    """(lat, lon, msl, u10, v10, z at LEVELS, lsm, time) of prediction `step` (1-based), float32."""
    lat = np.linspace(90, -90, 361)
    lon = np.linspace(0, 360, 720, endpoint=False)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    c_lat, c_lon = START[0] + 0.9 * step, (lon0 + 1.3 * step + 0.05 * step * step) % 360
    d_lon = (lo - c_lon + 180) % 360 - 180
    r2 = (la - c_lat) ** 2 + (d_lon * np.cos(np.deg2rad(c_lat))) ** 2
    depth = 0.0 if step in (9, 10) else 1.0           # the storm vanishes for two steps
    rng = np.random.default_rng(1000 + step)
    noise = depth if step != 10 else 1.0              # step 9: smooth tilted fields, no interior minimum anywhere -> a failure
    tilt = (1.0 - noise) * (3.0 * la + 2.0 * lo)
    msl = 101000.0 - depth * 4000.0 * np.exp(-r2 / 6.0) + noise * 40.0 * rng.standard_normal(la.shape) + tilt
    z700 = 30000.0 - depth * 900.0 * np.exp(-r2 / 14.0) + noise * 8.0 * rng.standard_normal(la.shape) + tilt
    ring = depth * 45.0 * np.sqrt(r2 / 2.0) * np.exp(0.5 - r2 / 4.0)
    ang = np.arctan2(la - c_lat, d_lon)
    u10 = -ring * np.sin(ang) + rng.standard_normal(la.shape)
    v10 = ring * np.cos(ang) + rng.standard_normal(la.shape)
    lsm = (((la - 17.5) ** 2 + (((lo - (lon0 + 8.5) + 180) % 360 - 180) ** 2)) < 6.0).astype(np.float64)   # an island in the path
    z = np.stack([z700 + 20000.0, z700, z700 - 15000.0])
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    return f(lat), f(lon), f(msl), f(u10), f(v10), f(z), f(lsm), START[2] + timedelta(hours=6 * step)


def batch(step: int, Batch, Metadata, device="cpu", lon0: float = START[1]):
    lat, lon, msl, u10, v10, z, lsm, time = fields(step, lon0)
    zero = torch.zeros_like(msl)
    return Batch(
        surf_vars={"2t": zero[None, None], "10u": u10[None, None], "10v": v10[None, None], "msl": msl[None, None]},
        static_vars={"lsm": lsm, "z": zero, "slt": zero},
        atmos_vars={"z": z[None, None]},
        metadata=Metadata(lat=lat, lon=lon, time=(time,), atmos_levels=LEVELS),
    ).to(device)
