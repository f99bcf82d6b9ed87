"""netCDF back end of `Batch.to_netcdf` / `Batch.from_netcdf` (reference: aurora/batch.py:224-292).

The reference writes through `xarray` (+ netCDF4) and raises when it is missing.  Here `xarray` is used when it is
installed -- the files are then exactly the reference's -- and otherwise the classic netCDF-3 writer that ships with
SciPy (`scipy.io.netcdf_file`, 64-bit offsets) produces a file with the same variables, dimensions and coordinates that
`xarray.load_dataset` / `open_mfdataset` decode to the same Dataset (CF time units on `time`).  Host memory only: callers
hand in CPU tensors (`rollout(..., to_host=True)` delivers them in pinned memory without stalling the device).
"""
from __future__ import annotations

from datetime import datetime, timezone
from pathlib import Path

import numpy as np

_EPOCH_UNITS = "seconds since 1970-01-01 00:00:00"


def have_xarray() -> bool:
    try:
        import xarray  # noqa: F401
    except ImportError:
        return False
    return True


def _seconds(t: datetime) -> float:
    if t.tzinfo is None:   # naive datetimes are UTC wall-clock, as numpy's datetime64 treats them
        t = t.replace(tzinfo=timezone.utc)
    return t.timestamp()


def write_dataset(path, data: dict, coords: dict, attrs: dict | None = None) -> None:
    """`data`: name -> (dims, ndarray); `coords`: latitude, longitude, time (datetimes), level, rollout_step (int)."""
    attrs = attrs or {}
    if have_xarray():
        import xarray as xr

        xr.Dataset(data, coords=coords, attrs=attrs).to_netcdf(path)
        return
    from scipy.io import netcdf_file

    sizes: dict[str, int] = {}
    for dims, arr in data.values():
        for d, n in zip(dims, arr.shape):
            assert sizes.setdefault(d, n) == n, f"dimension {d}: {sizes[d]} vs {n}"
    sizes.setdefault("latitude", len(coords["latitude"]))
    sizes.setdefault("longitude", len(coords["longitude"]))
    with netcdf_file(str(path), "w", version=2) as f:
        for d, n in sizes.items():
            f.createDimension(d, n)
        if "batch" in sizes:
            # `time` has one entry per batch element; xarray gives it its own dimension of that length
            f.createDimension("time", len(coords["time"]))
        if "level" in sizes:
            lv = f.createVariable("level", "d", ("level",))
            lv[:] = np.asarray(coords["level"], dtype=np.float64)
        for name in ("latitude", "longitude"):
            arr = np.asarray(coords[name])
            assert arr.ndim == 1, "the SciPy netCDF-3 writer takes vector coordinates (install xarray for matrices)"
            v = f.createVariable(name, "d" if arr.dtype == np.float64 else "f", (name,))
            v[:] = arr
        if "batch" in sizes:
            t = f.createVariable("time", "d", ("time",))
            t[:] = np.asarray([_seconds(x) for x in coords["time"]], dtype=np.float64)
            t.units = _EPOCH_UNITS
            t.calendar = "proleptic_gregorian"
        rs = f.createVariable("rollout_step", "i", ())
        rs.data[...] = int(coords["rollout_step"])   # (assignValue indexes a 0-d array with [:] under numpy 2)
        for name, (dims, arr) in data.items():
            arr = np.ascontiguousarray(arr)
            code = {"float32": "f", "float64": "d", "int32": "i"}[str(arr.dtype)]
            v = f.createVariable(name, code, dims)
            v[:] = arr
            v.coordinates = "rollout_step"
        for k, val in attrs.items():
            setattr(f, k, val)


def read_dataset(path) -> tuple[dict, dict, dict]:
    """-> (variables name -> ndarray, coords as in `write_dataset`, global attributes)."""
    if have_xarray():
        import xarray as xr

        ds = xr.load_dataset(path, engine="netcdf4")
        data = {k: ds[k].values for k in ds.data_vars}
        coords = {"latitude": ds.latitude.values, "longitude": ds.longitude.values,
                  "time": tuple(ds.time.values.astype("datetime64[s]").tolist()), "level": tuple(ds.level.values),
                  "rollout_step": int(ds.rollout_step.values)}
        return data, coords, dict(ds.attrs)
    from scipy.io import netcdf_file

    def native(a):   # netCDF-3 stores big-endian
        a = np.asarray(a)
        return a.astype(a.dtype.newbyteorder("="), copy=True)

    with netcdf_file(str(path), "r", mmap=False) as f:
        names = ("latitude", "longitude", "time", "level", "rollout_step")
        data = {k: native(v[:]) for k, v in f.variables.items() if k not in names}
        level = native(f.variables["level"][:]) if "level" in f.variables else np.zeros(0)
        level = tuple(int(x) if float(x).is_integer() else float(x) for x in level)
        times = tuple(datetime.fromtimestamp(float(s), tz=timezone.utc).replace(tzinfo=None)
                      for s in (f.variables["time"][:] if "time" in f.variables else ()))
        coords = {"latitude": native(f.variables["latitude"][:]), "longitude": native(f.variables["longitude"][:]),
                  "time": times, "level": level, "rollout_step": int(f.variables["rollout_step"].getValue())}
        attrs = {k: (v.decode() if isinstance(v, bytes) else native(v)) for k, v in f._attributes.items()}
    return data, coords, attrs


def band_paths(template) -> list[Path]:
    """Files written from one `path_template` by the ranks of a sharded forecast (`{rank}` -> any rank)."""
    import glob
    import re

    pattern = re.sub(r"\{rank[^}]*\}", "*", str(template))
    return sorted(Path(p) for p in glob.glob(pattern))
