"""The CPU oracle reproduces the golden vectors generated from the reference (fp64).

This is the pin that lets the oracle stand in for the reference on the GPU box, where
/root/reference does not exist.
"""
import pytest
import torch

from aurora_amd import normalisation
from oracle import aurora_oracle as oracle
from tests import helpers
from tests.golden_cases import CASES


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    case, model = helpers.case_model_meta(name)
    cfg = model.config
    sd = helpers.case_state_dict(model, torch.float64)
    surf, static, atmos, lat, lon, times = helpers.case_inputs(case, cfg)
    gold = helpers.load_golden(name)
    seen = 0
    with torch.inference_mode():
        gen = oracle.rollout(sd, cfg, surf, static, atmos, lat, lon, times, case["levels"],
                             case["steps"], normalisation.locations, normalisation.scales,
                             variant=model.variant)
        for s, (sp, ap, _) in enumerate(gen):
            for kind, d in (("surf", sp), ("atmos", ap)):
                for k, v in d.items():
                    ref = torch.from_numpy(gold[f"s{s}.{kind}.{k}"])
                    assert v.shape == ref.shape
                    # The golden is a float32 cast of the fp64 reference output (6e-8).  The bound also
                    # has to absorb the host CPU: lat/lon pooling and patch areas are fp32 upstream
                    # even in an fp64 model, and their last-bit differences between CPU families
                    # reach the output at the 1e-6 level (measured 2e-7 .. 3e-6 on EPYC vs Xeon).
                    v, ref, flipped = helpers.nan_agreement(v, ref)
                    assert flipped == 0 and helpers.rel_err(v, ref) < 2e-5, (name, s, kind, k)
                    seen += 1
    assert seen == len(gold)


def test_fp32_oracle_close_to_fp64():
    """Like-for-like fp32 run stays within the reference test's own tolerance form."""
    name = "base_pad"
    case, model = helpers.case_model_meta(name)
    cfg = model.config
    sd = helpers.case_state_dict(model, torch.float32)
    surf, static, atmos, lat, lon, times = helpers.case_inputs(case, cfg)
    gold = helpers.load_golden(name)
    with torch.inference_mode():
        sp, ap, _ = oracle.forward(sd, cfg, surf, static, atmos, lat, lon, times, case["levels"],
                                   0, normalisation.locations, normalisation.scales)
    for kind, d in (("surf", sp), ("atmos", ap)):
        for k, v in d.items():
            assert helpers.mean_rel_err(v, torch.from_numpy(gold[f"s0.{kind}.{k}"])) < 1e-4
