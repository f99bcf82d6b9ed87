"""`Batch` / `Metadata` host methods against the reference's (aurora/batch.py): the reference's own tests
(tests/test_batch.py: regridding to the same resolution is the identity, save / load round trip), values the reference
produced for `regrid`, `normalise`, `unnormalise`, `crop` on a seeded batch (tests/golden/batch_methods.npz, made by
tools/make_batch_golden.py), and the validation errors of batch.py:45-68, 142-168."""
from datetime import datetime
from pathlib import Path

import numpy as np
import pytest
import torch

from aurora_amd import Batch, Metadata

GOLD = Path(__file__).parent / "golden" / "batch_methods.npz"


def seeded_batch(B=Batch, M=Metadata, n_lat=17, n_lon=32, coord_dtype=torch.float32):
    g = torch.Generator().manual_seed(7)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    return B(
        surf_vars={k: r(1, 2, n_lat, n_lon) for k in ("2t", "10u", "10v", "msl")},
        static_vars={k: r(n_lat, n_lon) for k in ("lsm", "z", "slt")},
        atmos_vars={k: r(1, 2, 4, n_lat, n_lon) for k in ("z", "u", "v", "t", "q")},
        metadata=M(lat=torch.linspace(90, -90, n_lat, dtype=coord_dtype), lon=torch.linspace(0, 360, n_lon + 1, dtype=coord_dtype)[:-1],
                   time=(datetime(2020, 6, 1, 12, 0),), atmos_levels=(100, 250, 500, 850)),
    )


def test_interpolation_to_the_same_resolution_is_the_identity():
    """tests/test_batch.py:12-39 upstream (there on the 0.45-degree test data; here 401 x 800 points of the same spacing)."""
    b = seeded_batch(n_lat=401, n_lon=800, coord_dtype=torch.float64)   # float64 coordinates, as in the reference's test data
    rg = b.regrid(0.45).crop(4)      # regridding adds the south pole; remove it again
    b = b.crop(4)
    for k in b.surf_vars:
        np.testing.assert_allclose(b.surf_vars[k], rg.surf_vars[k], rtol=5e-6, atol=1e-6)
    for k in b.static_vars:
        np.testing.assert_allclose(b.static_vars[k], rg.static_vars[k], atol=1e-6)
    for k in b.atmos_vars:
        np.testing.assert_allclose(b.atmos_vars[k], rg.atmos_vars[k], rtol=5e-6, atol=1e-6)
    np.testing.assert_allclose(b.metadata.lat, rg.metadata.lat, atol=1e-5)
    np.testing.assert_allclose(b.metadata.lon, rg.metadata.lon, atol=1e-5)


def test_save_load(tmp_path):
    """tests/test_batch.py:42-60 upstream."""
    b = seeded_batch()
    b.to_netcdf(tmp_path / "batch.nc")
    loaded = Batch.from_netcdf(tmp_path / "batch.nc")
    for mine, theirs in ((b.surf_vars, loaded.surf_vars), (b.static_vars, loaded.static_vars), (b.atmos_vars, loaded.atmos_vars)):
        for k in mine:
            np.testing.assert_allclose(mine[k], theirs[k])
    np.testing.assert_allclose(b.metadata.lat, loaded.metadata.lat)
    np.testing.assert_allclose(b.metadata.lon, loaded.metadata.lon)
    assert b.metadata.time == loaded.metadata.time
    assert b.metadata.atmos_levels == loaded.metadata.atmos_levels
    assert b.metadata.rollout_step == loaded.metadata.rollout_step


def test_methods_equal_the_reference_on_a_seeded_batch():
    with np.load(GOLD) as z:
        gold = {k: z[k] for k in z.files}
    b = seeded_batch()
    rg = b.regrid(7.5)
    for grp, d in (("surf", rg.surf_vars), ("static", rg.static_vars), ("atmos", rg.atmos_vars)):
        for k, v in d.items():
            assert v.dtype == torch.float32
            np.testing.assert_allclose(v.numpy(), gold[f"regrid.{grp}.{k}"], rtol=1e-6, atol=1e-6, err_msg=k)
    np.testing.assert_array_equal(rg.metadata.lat.numpy(), gold["regrid.lat"])
    np.testing.assert_array_equal(rg.metadata.lon.numpy(), gold["regrid.lon"])
    stats = {"2t": (270.0, 30.0)}
    nb = b.normalise(surf_stats=stats)
    for grp, d in (("surf", nb.surf_vars), ("static", nb.static_vars), ("atmos", nb.atmos_vars)):
        for k, v in d.items():
            np.testing.assert_array_equal(v.numpy(), gold[f"normalise.{grp}.{k}"], err_msg=k)
    for k, v in nb.unnormalise(surf_stats=stats).surf_vars.items():
        np.testing.assert_array_equal(v.numpy(), gold[f"unnormalise.surf.{k}"], err_msg=k)
    cr = b.crop(4)
    np.testing.assert_array_equal(cr.metadata.lat.numpy(), gold["crop.lat"])
    np.testing.assert_array_equal(cr.surf_vars["2t"].numpy(), gold["crop.2t"])
    assert cr.spatial_shape == (16, 32) and b.crop(4).crop(4).spatial_shape == (16, 32)


def test_validation_errors_as_upstream():
    lat, lon = torch.linspace(90, -90, 17), torch.linspace(0, 360, 33)[:-1]
    ok = dict(time=(datetime(2020, 1, 1),), atmos_levels=(500,))
    for bad_lat, bad_lon, msg in (
        (lat * 1.1, lon, r"range \[-90, 90\]"), (lat, lon + 1.0 * (lon > 300) * 100, r"range \[0, 360\)"),
        (lat.flip(0), lon, "strictly decreasing"), (lat, lon.flip(0), "strictly increasing"),
        (lat[:, None].expand(17, 32), lon, "both be vectors or both be matrices"),
    ):
        with pytest.raises(ValueError, match=msg):
            Metadata(lat=bad_lat, lon=bad_lon, **ok)
    Metadata(lat=lat[:, None].expand(17, 32), lon=lon[None, :].expand(17, 32), **ok)   # matrices are fine
    b = seeded_batch()
    with pytest.raises(ValueError, match="multiple of the patch size"):
        b.crop(5)
    with pytest.raises(ValueError, match="at most be one latitude too many"):
        seeded_batch(n_lat=18).crop(4)
    with pytest.raises(KeyError):   # a variable without normalisation statistics (normalisation.py:44-45)
        Batch({"nope": b.surf_vars["2t"]}, b.static_vars, b.atmos_vars, b.metadata).normalise(surf_stats={})
