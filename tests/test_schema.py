"""state_dict key/shape schema equals the reference's for every public class."""
import gzip
import json

import pytest
import torch

import aurora_amd
from tests import helpers


@pytest.fixture(scope="module")
def schemas():
    with gzip.open(helpers.GOLD / "state_dict_schemas.json.gz", "rt") as f:
        return json.load(f)


@pytest.mark.parametrize("cls", ["Aurora", "AuroraPretrained", "AuroraSmallPretrained",
                                 "Aurora12hPretrained", "AuroraHighRes", "AuroraAirPollution",
                                 "AuroraWave"])
def test_state_dict_schema(cls, schemas):
    with torch.device("meta"):
        model = getattr(aurora_amd, cls)()
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert mine == schemas[cls]


def test_decoder_init_like_reference():
    """tests/test_model.py:113-123 upstream: head biases zero, weights non-zero."""
    model = aurora_amd.AuroraSmallPretrained()
    for k, v in model.state_dict().items():
        if k.startswith("decoder.surf_heads") or k.startswith("decoder.atmos_heads"):
            if k.endswith(".bias"):
                assert (v == 0).all()
            else:
                assert not (v == 0).all()
    assert aurora_amd.AuroraSmall is aurora_amd.AuroraSmallPretrained


def test_zero_inits_and_first_param():
    model = aurora_amd.AuroraSmallPretrained(use_lora=True)
    sd = model.state_dict()
    assert (sd["backbone.encoder_layers.0.blocks.0.attn.lora_qkv.loras.0.lora_B"] == 0).all()
    assert not (sd["backbone.encoder_layers.0.blocks.0.attn.lora_qkv.loras.0.lora_A"] == 0).all()
    assert (sd["backbone.encoder_layers.0.blocks.0.norm1.ln_modulation.1.weight"] == 0).all()
    assert (sd["encoder.surf_norm.weight"] == 1).all()
    assert next(model.parameters()).dtype == torch.float32


def test_cpu_forward_fails_loudly():
    model = aurora_amd.AuroraSmallPretrained()
    with pytest.raises(RuntimeError, match="HIP device"):
        model.forward(None)


def test_a_model_built_under_inference_mode_can_be_moved():
    """`Aurora._apply` stamps parameters and buffers (data_ptr, dtype, device, version) to decide whether the packed weights
    survive a `.to()`; inference tensors carry no version counter -- asking for one used to raise out of `.to()`."""
    with torch.inference_mode():
        model = aurora_amd.AuroraSmallPretrained()
    assert model.to("cpu") is model
    with torch.inference_mode():
        assert model.double() is model and next(model.parameters()).dtype == torch.float64
