"""Checkpoint adapters vs the reference's (aurora/model/compat.py:18-284, aurora/model/aurora.py:432-504).

tools/make_compat_fixtures.py ran the REFERENCE adapters on synthetic published-layout checkpoints
(tests/compat_recipes.py) and stored a digest (shape + CRC-32 of the bytes) per adapted tensor.  Here the same inputs go
through aurora_amd/model/compat.py via the model classes' `_adapt_checkpoint`; key sets and every tensor must agree
exactly, and the result must load with `strict=True` -- mirroring tests/test_checkpoint_adaptation.py:24-60 upstream,
extended to the variant adapters, which the reference does not test at all.
"""
import gzip
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import aurora_amd
from tests.compat_recipes import FAMILIES, digest, old_layout

with gzip.open(Path(__file__).parent / "golden" / "compat_fixtures.json.gz", "rt") as f:
    FIX = json.load(f)


def _model(fam, **extra):
    spec = FAMILIES[fam]
    return getattr(aurora_amd, spec["cls"])(**dict(spec["kwargs"], **extra)), spec


@pytest.mark.parametrize("fam", list(FAMILIES))
def test_adapter_matches_reference(fam):
    model, spec = _model(fam)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    old = old_layout(fam, shapes, spec["patch"])
    adapted = model._adapt_checkpoint({k: v.clone() for k, v in old.items()})
    mine, ref = digest(adapted), FIX[fam]
    assert set(mine) == set(ref), (sorted(set(mine) - set(ref))[:5], sorted(set(ref) - set(mine))[:5])
    bad = [k for k in ref if mine[k] != ref[k]]
    assert not bad, bad[:10]
    model.load_state_dict(adapted, strict=True)          # complete and shape-correct
    for k, v in model.state_dict().items():
        assert torch.equal(v, adapted[k]), k


def test_adapt_is_idempotent_on_current_layout():
    """A state dict already in the current layout passes through every adapter unchanged."""
    for fam in FAMILIES:
        model, _ = _model(fam)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        if fam == "air_pollution":   # the indexing-bug emulation aliases z to static_z (compat.py:156-159 upstream)
            continue
        out = model._adapt_checkpoint(dict(sd))
        assert set(out) == set(sd)
        for k in sd:
            assert torch.equal(out[k], sd[k]), k


# ---- history extension: tests/test_checkpoint_adaptation.py upstream, same assertions --------------------------------
@pytest.fixture
def checkpoint():
    return {
        "encoder.surf_token_embeds.weights.0": torch.rand((2, 1, 2, 4, 4)),
        "encoder.atmos_token_embeds.weights.0": torch.rand((2, 1, 2, 4, 4)),
    }


@pytest.mark.parametrize("hist", [4, 5])
def test_adapt_checkpoint_max_history(hist, checkpoint):
    model = aurora_amd.AuroraSmallPretrained(max_history_size=hist)
    before = {k: v.clone() for k, v in checkpoint.items()}
    assert checkpoint["encoder.surf_token_embeds.weights.0"].shape[2] == 2
    model.adapt_checkpoint_max_history_size(checkpoint)
    for name, weight in checkpoint.items():
        assert weight.shape[2] == model.max_history_size
        for j in range(weight.shape[2]):
            if j >= before[name].shape[2]:
                np.testing.assert_allclose(weight[:, :, j], 0 * weight[:, :, j])
            else:
                np.testing.assert_allclose(weight[:, :, j], before[name][:, :, j])


def test_adapt_checkpoint_max_history_fail(checkpoint):
    model = aurora_amd.AuroraSmallPretrained(max_history_size=1)
    with pytest.raises(AssertionError):
        model.adapt_checkpoint_max_history_size(checkpoint)


def test_adapt_checkpoint_max_history_twice(checkpoint):
    model = aurora_amd.AuroraSmallPretrained(max_history_size=4)
    model.adapt_checkpoint_max_history_size(checkpoint)
    once = {k: v.clone() for k, v in checkpoint.items()}
    model.adapt_checkpoint_max_history_size(checkpoint)
    for k in once:
        assert torch.equal(once[k], checkpoint[k])


def test_history_extension_matches_reference_and_loads(tmp_path):
    """Published layout -> adapters -> history 2 -> 5, through `load_checkpoint_local` (the whole load path)."""
    model, spec = _model("pretrained", max_history_size=5)
    base, _ = _model("pretrained")
    shapes = {k: tuple(v.shape) for k, v in base.state_dict().items()}
    old = old_layout("pretrained", shapes, spec["patch"])
    path = tmp_path / "published.ckpt"
    torch.save(old, path)
    model.load_checkpoint_local(str(path), strict=True)
    got = digest({k: v for k, v in model.state_dict().items() if "token_embeds.weights" in k})
    assert got == FIX["history5"]
