"""Shared test plumbing: build (config, weights, inputs) for a golden case."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

import aurora_amd
from aurora_amd import normalisation
from aurora_amd.model.schema import param_specs
from oracle import detdata
from tests.golden_cases import CASES

GOLD = Path(__file__).parent / "golden"


def case_model_meta(name: str):
    case = CASES[name]
    with torch.device("meta"):
        model = getattr(aurora_amd, case["cls"])(**case["kwargs"])
    return case, model


def case_state_dict(model, dtype=torch.float64):
    """Deterministic weights for every entry of the model's state_dict."""
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    return detdata.det_state_dict(shapes, dtype)


def case_inputs(case, cfg):
    if case["cls"] == "AuroraWave":
        return detdata.det_wave_inputs(cfg.static_vars, cfg.atmos_vars, case["B"], case["T"], case["H"], case["W"],
                                       case["levels"], normalisation.locations, normalisation.scales)
    return detdata.det_inputs(
        cfg.surf_vars, cfg.static_vars, cfg.atmos_vars, case["B"], case["T"], case["H"], case["W"],
        case["levels"], normalisation.locations, normalisation.scales,
        positive=cfg.positive_surf_vars + cfg.positive_atmos_vars,
    )


def load_golden(name: str) -> dict[str, np.ndarray]:
    with np.load(GOLD / f"{name}.npz") as z:
        return {k: z[k] for k in z.files}


def nan_agreement(a: torch.Tensor, b: torch.Tensor):
    """(a', b', fraction of points where exactly one of a, b is NaN): the points where both are
    finite, for value comparison, and how often the NaN masks disagree (ocean-wave outputs carry NaN
    for absent wave systems; a density logit within rounding of 0 may legitimately flip)."""
    na, nb = torch.isnan(a), torch.isnan(b)
    both = ~na & ~nb
    return a[both], b[both], (na ^ nb).double().mean().item()


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a - b| / max |b|"""
    return (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-30)


def mean_rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """The reference test's metric (tests/test_model.py:45-61): mean|a-b| / mean|b|."""
    return (a.double() - b.double()).abs().mean().item() / (b.double().abs().mean().item() + 1e-30)
