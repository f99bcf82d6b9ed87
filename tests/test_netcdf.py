"""`Batch.to_netcdf` / `from_netcdf` (reference aurora/batch.py:224-292) and the sharded output path (SURVEY.md section 8
f-4): every rank of a latitude-band forecast writes its own file, nothing is gathered, and the files reassemble to
exactly what the un-sharded `to_netcdf` writes.  Runs on CPU: two gloo processes play the ranks."""
import os
from datetime import datetime

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aurora_amd import Batch, Metadata
from aurora_amd.batch import BandBatch, derive_metadata
from aurora_amd.engine import geometry, partition

LEVELS = (100, 250, 500, 850)
P = 4


def _batch(H=192, W=32, seed=0):   # 48 patch rows: 12 at the coarsest stage, enough for three bands
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    return Batch({k: r(1, 1, H, W) for k in ("2t", "msl")}, {k: r(H, W) for k in ("lsm", "z")},
                 {k: r(1, 1, len(LEVELS), H, W) for k in ("t", "q")},
                 Metadata(torch.linspace(89, -89, H), torch.linspace(0, 360, W + 1)[:-1],
                          (datetime(2020, 6, 1, 12, 0),), LEVELS, rollout_step=3))


def _assert_same(a: Batch, b: Batch):
    for da, db in ((a.surf_vars, b.surf_vars), (a.static_vars, b.static_vars), (a.atmos_vars, b.atmos_vars)):
        assert list(da) == list(db)
        for k in da:
            assert da[k].dtype == db[k].dtype and torch.equal(da[k], db[k]), k
    assert torch.equal(a.metadata.lat, b.metadata.lat) and torch.equal(a.metadata.lon, b.metadata.lon)
    assert tuple(a.metadata.time) == tuple(b.metadata.time)
    assert tuple(a.metadata.atmos_levels) == tuple(b.metadata.atmos_levels)
    assert a.metadata.rollout_step == b.metadata.rollout_step


def test_netcdf_round_trip(tmp_path):
    b = _batch()
    b.to_netcdf(tmp_path / "full.nc")
    _assert_same(Batch.from_netcdf(tmp_path / "full.nc"), b)


def test_band_needs_a_rank_field(tmp_path):
    b = _batch()
    band = BandBatch(b.surf_vars, b.static_vars, b.atmos_vars, b.metadata, full_patch_rows=48, band=(0, 48))
    with pytest.raises(ValueError, match="rank"):
        band.to_netcdf(tmp_path / "x.nc")


def _band_of(b: Batch, rank: int, world: int) -> BandBatch:
    """What `Engine.local_band` hands a rank (aurora_amd/engine/engine.py), without the HIP library."""
    H, W = b.spatial_shape
    all_res, _ = geometry.stage_resolutions((4, H // P, W // P), 3)
    h0, h1 = partition.band_rows(all_res, (2, 6, 12), world)[0][rank]
    cut = lambda d: {k: v[..., h0 * P:h1 * P, :] for k, v in d.items()}  # noqa: E731
    return BandBatch(cut(b.surf_vars), cut(b.static_vars), cut(b.atmos_vars),
                     derive_metadata(b.metadata, lat=b.metadata.lat[h0 * P:h1 * P]), full_patch_rows=H // P,
                     band=(h0, h1), rank=rank, world=world)


def _rank_main(rank: int, world: int, port: int, tmp: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = _batch()
        band = _band_of(full, rank, world)
        band.to_netcdf(os.path.join(tmp, "pred.{rank:02d}.nc"))     # no collective: every rank writes its own rows
        dist.barrier()
        if rank == 0:
            full.to_netcdf(os.path.join(tmp, "full.nc"))
            joined = Batch.from_netcdf(os.path.join(tmp, "pred.{rank:02d}.nc"))
            _assert_same(joined, Batch.from_netcdf(os.path.join(tmp, "full.nc")))
            _assert_same(joined, full)
            assert sorted(os.listdir(tmp)) == ["full.nc", "pred.00.nc", "pred.01.nc"]
    finally:
        dist.destroy_process_group()


def test_two_ranks_write_their_bands_and_the_files_reassemble(tmp_path):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_band_files_carry_their_row_range(tmp_path):
    from aurora_amd import netcdf

    full = _batch()
    for r in range(3):
        _band_of(full, r, 3).to_netcdf(tmp_path / "p.{rank}.nc")
    rows = []
    for p in netcdf.band_paths(tmp_path / "p.{rank}.nc"):
        _, coords, attrs = netcdf.read_dataset(p)
        rows.append(tuple(int(x) for x in np.atleast_1d(attrs["aurora_band_patch_rows"])))
        assert int(attrs["aurora_full_patch_rows"]) == 48 and int(attrs["aurora_world"]) == 3
        assert len(coords["latitude"]) == (rows[-1][1] - rows[-1][0]) * P
    assert sorted(rows)[0][0] == 0 and sorted(rows)[-1][1] == 48
    _assert_same(Batch.from_netcdf([str(p) for p in netcdf.band_paths(tmp_path / "p.{rank}.nc")]), full)


def test_fill_keeps_unknown_fields():
    from aurora_amd.rollout import _fill

    assert _fill("out/s{step:03d}.r{rank:02d}.nc", step=7) == "out/s007.r{rank:02d}.nc"
    assert _fill("s{step}.nc", step=12) == "s12.nc"
    assert _fill("s{step}.r{rank}.nc", step=1).format(rank=3) == "s1.r3.nc"
