"""Host-side hooks of the ocean-wave variant against the oracle's restatement of the reference (CPU only)."""
import torch

from aurora_amd import Batch, Metadata, normalisation
from aurora_amd.model import wave
from oracle import aurora_oracle as oracle
from oracle import detdata


def _raw_batch(rollout_step=0):
    surf, static, atmos, lat, lon, times = detdata.det_wave_inputs(
        ("lsm", "z", "slt", "wmb", "lat_mask"), ("z", "u", "v", "t", "q"), 1, 2, 8, 16, (100, 250),
        normalisation.locations, normalisation.scales)
    f = lambda d: {k: v.float() for k, v in d.items()}  # noqa: E731
    md = Metadata(lat.float(), lon.float(), times, (100, 250), rollout_step=rollout_step)
    return Batch(f(surf), f(static), f(atmos), md)


def test_batch_transform_matches_the_reference_restatement():
    batch = _raw_batch()
    got = wave.transform_batch(batch).surf_vars
    want = oracle.wave_batch_transform(batch.surf_vars, 0)
    assert tuple(got) == tuple(want) and "dwi" not in got and "10u_wave" in got
    for k in want:
        assert torch.equal(torch.isnan(got[k]), torch.isnan(want[k])), k
        assert torch.equal(got[k].nan_to_num(0.0), want[k].nan_to_num(0.0)), k
    assert torch.isnan(got["swh"]).any() and torch.isnan(got["mwd"]).any()
    assert not torch.isnan(got["mpts"]).any()          # the reference lists mdts twice, so mpts is never masked


def test_batch_transform_is_idempotent_and_skips_marking_after_step_0():
    batch = _raw_batch()
    once = wave.transform_batch(batch)
    twice = wave.transform_batch(once)
    for k, v in once.surf_vars.items():
        assert torch.equal(v.nan_to_num(-7.0), twice.surf_vars[k].nan_to_num(-7.0)), k
    assert "dwi" in batch.surf_vars                    # the input batch is not mutated
    later = wave.transform_batch(_raw_batch(rollout_step=3))
    assert not torch.isnan(later.surf_vars["swh"]).any()   # zero heights are only marked on analysis data
