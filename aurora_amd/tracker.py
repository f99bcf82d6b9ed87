"""Tropical-cyclone tracker over predictions that stay on the device (reference: aurora/tracker.py:126-282).

The reference moves every prediction to the CPU (`batch.to("cpu")`, 286 MB per step at 0.25 degree) and then looks at a
few 10-degree windows of four fields.  Here the windows are cut out ON the device (two `index_select`s per field) and only
they travel -- some tens of kilobytes per step -- so a roll-out can be tracked without ever leaving the HBM; what is done
with a window afterwards is the reference's arithmetic on the same float32 values, in numpy / scipy on the host (Gaussian
smoothing, 8 x 8 minimum filter, nearest local minimum by great-circle distance), so tracks are equal to the reference's
(tests/test_tracker.py holds a track the reference produced).

Search order per step, as upstream (`:160-282`): extrapolate the last eight fixes linearly; snap to the nearest
mean-sea-level-pressure minimum in boxes of 5, 4, 3, 2, 1.5 degrees whose surroundings are free of land; failing that
to the nearest 700 hPa geopotential minimum (then refine on pressure); failing that keep the extrapolated guess and count
a failure.  Minimum pressure and maximum 10 m wind come from a 1.5-degree crop around the fix.
"""
from __future__ import annotations

from datetime import datetime

import numpy as np
import torch

from aurora_amd.batch import Batch

__all__ = ["Tracker", "NoEyeException"]

EARTH_RADIUS_KM = 6371
BOX_DEGREES = (5, 4, 3, 2, 1.5)   # search boxes for the pressure minimum, widest first
CROP_DEGREES = 1.5                # window of the reported minimum pressure / maximum wind
FIT_POINTS = 8                    # fixes the linear extrapolation looks back at


class NoEyeException(Exception):
    """No eye can be found."""


class _Grid:
    """Latitude / longitude axes on the host, and windows of device fields cut by index."""

    def __init__(self, lat: torch.Tensor, lon: torch.Tensor) -> None:
        self.lat, self.lon = lat.cpu().numpy(), lon.cpu().numpy()

    def window_indices(self, lat0: float, lat1: float, lon0: float, lon1: float) -> tuple[np.ndarray, np.ndarray]:
        rows = np.flatnonzero((lat0 <= self.lat) & (self.lat <= lat1))
        lon0, lon1 = lon0 % 360, lon1 % 360
        if lon0 <= lon1:
            cols = np.flatnonzero((lon0 <= self.lon) & (self.lon <= lon1))
        else:   # the window straddles the date line: the eastern piece first, then the western
            cols = np.concatenate((np.flatnonzero(lon0 <= self.lon), np.flatnonzero(self.lon <= lon1)))
        return rows, cols

    def cut(self, field: torch.Tensor, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
        """field[..., rows, cols] as a host array; only the window crosses the bus."""
        r = torch.as_tensor(rows, device=field.device)
        c = torch.as_tensor(cols, device=field.device)
        return field.index_select(-2, r).index_select(-1, c).cpu().numpy()

    def box(self, field: torch.Tensor, lat: float, lon: float, d_lat: float, d_lon: float):
        rows, cols = self.window_indices(lat - d_lat, lat + d_lat, lon - d_lon, lon + d_lon)
        return self.lat[rows], self.lon[cols], self.cut(field, rows, cols)


def great_circle_km(lat1, lon1, lat2, lon2):
    """Haversine distance (in the reference's algebraic form, tracker.py:52-58, so that ties break alike)."""
    lat1, lat2, lon1, lon2 = np.deg2rad(lat1), np.deg2rad(lat2), np.deg2rad(lon1), np.deg2rad(lon2)
    inner = 1 - np.cos(lat2 - lat1) + np.cos(lat1) * np.cos(lat2) * (1 - np.cos(lon2 - lon1))
    return 2 * EARTH_RADIUS_KM * np.arcsin(np.sqrt(0.5 * inner))


def nearest_minimum(lats: np.ndarray, lons: np.ndarray, box: np.ndarray, lat: float, lon: float, cap: int = 8) -> tuple[float, float]:
    """The local minimum of the smoothed window closest to (lat, lon); minima on the window's rim do not count."""
    from scipy.ndimage import gaussian_filter, minimum_filter

    smooth = gaussian_filter(box, sigma=1)
    is_min = minimum_filter(smooth, size=(cap, cap)) == smooth
    is_min[[0, -1], :] = False
    is_min[:, [0, -1]] = False
    ii, jj = np.nonzero(is_min)
    if ii.size == 0:
        raise NoEyeException()
    k = int(np.argmin(great_circle_km(lats[ii], lons[jj], lat, lon)))
    return lats[ii[k]], lons[jj[k]]


def extrapolate(lats: list[float], lons: list[float]) -> tuple[float, float]:
    """First guess of the next fix: a straight line through the last eight."""
    if len(lats) != len(lons):
        raise AssertionError("as many latitudes as longitudes")
    if not lats:
        raise ValueError("Cannot extrapolate from empty lists.")
    if len(lats) == 1:
        return lats[0], lons[0]
    track = np.stack((lats[-FIT_POINTS:], lons[-FIT_POINTS:]), axis=-1)
    line = np.polyfit(np.arange(len(track)), track, 1)
    return tuple(np.polyval(line, len(track)))


class Tracker:
    """Simple tropical cyclone tracker (same interface and results as `aurora.Tracker`)."""

    def __init__(self, init_lat: float, init_lon: float, init_time: datetime) -> None:
        self.tracked_times: list[datetime] = [init_time]
        self.tracked_lats: list[float] = [init_lat]
        self.tracked_lons: list[float] = [init_lon]
        self.tracked_msls: list[float] = [np.nan]
        self.tracked_winds: list[float] = [np.nan]
        self.fails: int = 0

    def results(self):
        """The track as a DataFrame: time, lat, lon, msl, wind."""
        import pandas as pd

        return pd.DataFrame({"time": self.tracked_times, "lat": self.tracked_lats, "lon": self.tracked_lons,
                             "msl": self.tracked_msls, "wind": self.tracked_winds})

    def step(self, batch: Batch) -> None:
        """Track the next step of a roll-out; `batch` may live on any device."""
        if len(batch.metadata.time) != 1:
            raise RuntimeError("Predictions don't have batch size one.")
        grid = _Grid(batch.metadata.lat, batch.metadata.lon)
        level = list(batch.metadata.atmos_levels).index(700)
        z700 = batch.atmos_vars["z"][0, 0, level]
        msl, u10, v10 = (batch.surf_vars[k][0, 0] for k in ("msl", "10u", "10v"))
        lsm = batch.static_vars["lsm"]
        when = batch.metadata.time[0]

        lat, lon = extrapolate(self.tracked_lats, self.tracked_lons)
        lat, lon = max(min(lat, 90), -90), lon % 360

        def over_water(lat: float, lon: float, delta: float) -> bool:
            return grid.box(lsm, lat, lon, delta, delta)[2].max() < 0.5

        def snap_to_pressure(lat: float, lon: float):
            """The nearest pressure minimum in the widest land-free box that has one, or None."""
            for delta in BOX_DEGREES:
                try:
                    if over_water(lat, lon, delta):
                        return nearest_minimum(*grid.box(msl, lat, lon, delta, delta), lat, lon)
                except NoEyeException:
                    pass
            return None

        fix = snap_to_pressure(lat, lon)
        if fix is None:
            try:   # the mid-troposphere vortex is broader and survives landfall: find it, then refine on pressure
                fix = nearest_minimum(*grid.box(z700, lat, lon, 5, 5), lat, lon)
                fix = snap_to_pressure(*fix) or fix
            except NoEyeException:
                pass
        if fix is None:
            self.fails += 1
            if len(self.tracked_lats) <= 1:
                raise NoEyeException("Completely failed at the first step.")
        else:
            lat, lon = fix

        self.tracked_times.append(when)
        self.tracked_lats.append(lat)
        self.tracked_lons.append(lon)
        rows, cols = grid.window_indices(lat - CROP_DEGREES, lat + CROP_DEGREES, lon - CROP_DEGREES, lon + CROP_DEGREES)
        u, v = grid.cut(u10, rows, cols), grid.cut(v10, rows, cols)
        self.tracked_msls.append(grid.cut(msl, rows, cols).min())
        self.tracked_winds.append(np.sqrt(u * u + v * v).max())
