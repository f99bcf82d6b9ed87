"""`Batch` and `Metadata`: the data containers of the public API.

Host-side mirror of the reference's `aurora/batch.py` (Metadata :23-68, Batch :71-292,
interpolate :299-362): same field names, constructor signature, validation errors and
method set, so user code written against the reference runs unchanged.
"""

from __future__ import annotations

import dataclasses
from datetime import datetime
from pathlib import Path
from typing import Callable

import numpy as np
import torch

from aurora_amd.normalisation import normalise_atmos_var, normalise_surf_var

__all__ = ["Metadata", "Batch", "BandBatch"]


@dataclasses.dataclass
class Metadata:
    """Coordinates and bookkeeping of a batch.

    lat / lon are vectors (strictly decreasing / strictly increasing) or matrices;
    `time` has one entry per batch element; `rollout_step` counts how many model steps
    produced this batch (0 = analysis data) and selects the LoRA set of the next step.
    """

    lat: torch.Tensor
    lon: torch.Tensor
    time: tuple[datetime, ...]
    atmos_levels: tuple[int | float, ...]
    rollout_step: int = 0

    def __post_init__(self) -> None:
        lat, lon = self.lat, self.lon
        if bool((lat > 90).any()) or bool((lat < -90).any()):
            raise ValueError("Latitudes must be in the range [-90, 90].")
        if bool((lon < 0).any()) or bool((lon >= 360).any()):
            raise ValueError("Longitudes must be in the range [0, 360).")

        if lat.dim() == 1 and lon.dim() == 1:
            if not bool((lat[1:] < lat[:-1]).all()):
                raise ValueError("Latitudes must be strictly decreasing.")
            if not bool((lon[1:] > lon[:-1]).all()):
                raise ValueError("Longitudes must be strictly increasing.")
        elif lat.dim() == 2 and lon.dim() == 2:
            # Same (lenient) test as the reference for matrices: the differences along a
            # column must all be non-zero (batch.py:58-59 there).
            if not bool(torch.all(lat[1:, :] - lat[:-1, :])):
                raise ValueError("Latitudes must be strictly decreasing along every column.")
            if not bool((lon[:, 1:] > lon[:, :-1]).all()):
                raise ValueError("Longitudes must be strictly increasing along every row.")
        else:
            raise ValueError(
                "The latitudes and longitudes must either both be vectors or both be matrices."
            )


def derive_metadata(md: Metadata, **changes) -> Metadata:
    """`dataclasses.replace` for metadata whose coordinates were validated already.

    Validation reads the coordinates on the host; for device-resident batches that is a
    device->host synchronisation per check.  Moving, casting or slicing valid coordinates keeps them
    valid, so the per-step paths (and hipGraph capture, where such a sync is illegal) derive new
    metadata without re-validating.
    """
    new = object.__new__(Metadata)
    for f in dataclasses.fields(Metadata):
        object.__setattr__(new, f.name, changes.get(f.name, getattr(md, f.name)))
    return new


@dataclasses.dataclass
class Batch:
    """A batch of data.

    surf_vars:   name -> (b, t, h, w)
    static_vars: name -> (h, w)
    atmos_vars:  name -> (b, t, c, h, w)
    """

    surf_vars: dict[str, torch.Tensor]
    static_vars: dict[str, torch.Tensor]
    atmos_vars: dict[str, torch.Tensor]
    metadata: Metadata

    @property
    def spatial_shape(self) -> tuple[int, int]:
        return tuple(next(iter(self.surf_vars.values())).shape[-2:])

    # -- elementwise utilities ------------------------------------------------------------
    def _affine(self, surf_stats, unnormalise: bool) -> "Batch":
        lv = self.metadata.atmos_levels
        return dataclasses.replace(
            self,
            surf_vars={
                k: normalise_surf_var(v, k, surf_stats, unnormalise)
                for k, v in self.surf_vars.items()
            },
            static_vars={
                k: normalise_surf_var(v, k, surf_stats, unnormalise)
                for k, v in self.static_vars.items()
            },
            atmos_vars={
                k: normalise_atmos_var(v, k, lv, unnormalise) for k, v in self.atmos_vars.items()
            },
        )

    def normalise(self, surf_stats: dict[str, tuple[float, float]]) -> "Batch":
        """(x - location) / scale for every variable; `surf_stats` overrides surface stats."""
        return self._affine(surf_stats, unnormalise=False)

    def unnormalise(self, surf_stats: dict[str, tuple[float, float]]) -> "Batch":
        """x * scale + location for every variable."""
        return self._affine(surf_stats, unnormalise=True)

    def crop(self, patch_size: int) -> "Batch":
        """Drop the last latitude row when there is exactly one row too many."""
        h, w = self.spatial_shape
        if w % patch_size != 0:
            raise ValueError("Width of the data must be a multiple of the patch size.")
        extra = h % patch_size
        if extra == 0:
            return self
        if extra != 1:
            raise ValueError(
                f"There can at most be one latitude too many, but there are {extra} too many."
            )
        cut = lambda d: {k: v[..., :-1, :] for k, v in d.items()}  # noqa: E731
        md = derive_metadata(self.metadata, lat=self.metadata.lat[:-1])
        return dataclasses.replace(self, surf_vars=cut(self.surf_vars), static_vars=cut(self.static_vars),
                                   atmos_vars=cut(self.atmos_vars), metadata=md)

    def _fmap(self, f: Callable[[torch.Tensor], torch.Tensor]) -> "Batch":
        md = derive_metadata(self.metadata, lat=f(self.metadata.lat), lon=f(self.metadata.lon))
        return dataclasses.replace(
            self,
            surf_vars={k: f(v) for k, v in self.surf_vars.items()},
            static_vars={k: f(v) for k, v in self.static_vars.items()},
            atmos_vars={k: f(v) for k, v in self.atmos_vars.items()},
            metadata=md,
        )

    def to(self, device: str | torch.device) -> "Batch":
        return self._fmap(lambda x: x.to(device))

    def type(self, t: type) -> "Batch":
        return self._fmap(lambda x: x.type(t))

    # -- regridding and I/O (not on the per-step path) -----------------------------------
    def regrid(self, res: float) -> "Batch":
        """Bilinear regrid to a `res`-degree grid (float32, CPU), periodic in longitude."""
        n_lat, n_lon = round(180 / res) + 1, round(360 / res)
        lat_new = torch.from_numpy(np.linspace(90, -90, n_lat))
        lon_new = torch.from_numpy(np.linspace(0, 360, n_lon, endpoint=False))
        lat, lon = self.metadata.lat, self.metadata.lon

        def f(v: torch.Tensor) -> torch.Tensor:
            return _interpolate(v, lat, lon, lat_new, lon_new)

        md = dataclasses.replace(self.metadata, lat=lat_new, lon=lon_new)
        return Batch(
            {k: f(v) for k, v in self.surf_vars.items()},
            {k: f(v) for k, v in self.static_vars.items()},
            {k: f(v) for k, v in self.atmos_vars.items()},
            md,
        )

    def _netcdf_payload(self):
        arr = lambda x: x.detach().cpu().numpy()  # noqa: E731
        data = {}
        for k, v in self.surf_vars.items():
            data[f"surf_{k}"] = (("batch", "history", "latitude", "longitude"), arr(v))
        for k, v in self.static_vars.items():
            data[f"static_{k}"] = (("latitude", "longitude"), arr(v))
        for k, v in self.atmos_vars.items():
            data[f"atmos_{k}"] = (("batch", "history", "level", "latitude", "longitude"), arr(v))
        coords = {
            "latitude": arr(self.metadata.lat),
            "longitude": arr(self.metadata.lon),
            "time": list(self.metadata.time),
            "level": list(self.metadata.atmos_levels),
            "rollout_step": self.metadata.rollout_step,
        }
        return data, coords

    def to_netcdf(self, path: str | Path) -> None:
        """Write the batch to a file (reference batch.py:224-257): through `xarray` when installed, else through
        SciPy's netCDF-3 writer with the same variables, dimensions and coordinates (aurora_amd/netcdf.py)."""
        from aurora_amd import netcdf

        data, coords = self._netcdf_payload()
        netcdf.write_dataset(path, data, coords)

    @classmethod
    def from_netcdf(cls, path) -> "Batch":
        """Load a batch from a file -- or from the per-rank files of a sharded forecast (`BandBatch.to_netcdf`): a
        list of paths, or the `{rank}` template they were written with; the latitude bands are joined north to south."""
        from aurora_amd import netcdf

        if isinstance(path, (list, tuple)):
            paths = [Path(p) for p in path]
        elif "{rank" in str(path):
            paths = netcdf.band_paths(path)
            if not paths:
                raise FileNotFoundError(f"no files match {path}")
        else:
            paths = [Path(path)]
        parts = [netcdf.read_dataset(p) for p in paths]
        parts.sort(key=lambda part: -float(part[1]["latitude"][0]))   # latitudes decrease: northernmost band first
        coords = parts[0][1]
        lat = np.concatenate([p_[1]["latitude"] for p_ in parts])
        groups: dict[str, dict[str, torch.Tensor]] = {"surf_": {}, "static_": {}, "atmos_": {}}
        for key in parts[0][0]:
            for prefix, dst in groups.items():
                if key.startswith(prefix):
                    dst[key[len(prefix):]] = torch.from_numpy(np.concatenate([p_[0][key] for p_ in parts], axis=-2))
                    break
        return cls(
            groups["surf_"],
            groups["static_"],
            groups["atmos_"],
            Metadata(
                lat=torch.from_numpy(lat),
                lon=torch.from_numpy(coords["longitude"]),
                time=tuple(coords["time"]),
                atmos_levels=tuple(coords["level"]),
                rollout_step=int(coords["rollout_step"]),
            ),
        )


@dataclasses.dataclass
class BandBatch(Batch):
    """A latitude band of a batch, as held by one rank of a sharded model (not in the reference).

    Produced by a sharded `Aurora.forward(..., gather_output=False)` and accepted back as input, so that
    a roll-out can stay distributed between steps.  `full_patch_rows` is the number of patch rows of the
    whole grid; `band` the owned patch rows `[h0, h1)`.
    """

    full_patch_rows: int = 0
    band: tuple[int, int] = (0, 0)
    rank: int = 0
    world: int = 1

    def crop(self, patch_size: int) -> "BandBatch":
        return self  # a band is cut out of an already cropped grid

    def to_netcdf(self, path: str | Path) -> None:
        """Sharded output: every rank writes ITS latitude band to its own file -- `path` is a
        template with a `{rank}` field, e.g. "step{rank:02d}.nc" -- and nothing is gathered.  Each file is a complete
        dataset on its latitude slice (same variables / dimensions / coordinates as `Batch.to_netcdf`, plus the band
        bookkeeping as global attributes), so `Batch.from_netcdf(template)` here, or
        `xarray.open_mfdataset(files, combine="by_coords")` anywhere, reassembles the un-sharded fields."""
        from aurora_amd import netcdf

        if "{rank" not in str(path):
            raise ValueError("a latitude band is written per rank: the path needs a `{rank}` field, e.g. 'pred.{rank}.nc'")
        data, coords = self._netcdf_payload()
        attrs = {"aurora_band_patch_rows": np.asarray(self.band, dtype=np.int32),
                 "aurora_full_patch_rows": np.int32(self.full_patch_rows),
                 "aurora_rank": np.int32(self.rank), "aurora_world": np.int32(self.world)}
        netcdf.write_dataset(str(path).format(rank=self.rank), data, coords, attrs)


def _interpolate(v, lat, lon, lat_new, lon_new) -> torch.Tensor:
    """Linear interpolation on a lat/lon grid in fp64; longitudes wrap, latitudes extrapolate."""
    from scipy.interpolate import RegularGridInterpolator

    lat64, lon64 = lat.double().numpy(), lon.double().numpy()
    assert (np.diff(lon64) > 0).all()
    lon_ext = np.concatenate((lon64[-1:] - 360, lon64, lon64[:1] + 360))
    fields = v.double().numpy()
    lead = fields.shape[:-2]
    fields = fields.reshape(-1, *fields.shape[-2:])
    targets = np.meshgrid(
        lat_new.double().numpy(), lon_new.double().numpy(), indexing="ij", sparse=True
    )
    out = np.empty((fields.shape[0], lat_new.shape[0], lon_new.shape[0]))
    for i, f in enumerate(fields):
        f_ext = np.concatenate((f[:, -1:], f, f[:, :1]), axis=1)
        rgi = RegularGridInterpolator(
            (lat64, lon_ext), f_ext, method="linear", bounds_error=False, fill_value=None
        )
        out[i] = rgi(tuple(targets))
    return torch.from_numpy(out.reshape(*lead, *out.shape[-2:])).float()
