"""aurora_amd.foundry: the reference's `Model.run` serving loop (aurora/foundry/common/model.py:21-150) over this package's
roll-out -- registry names as upstream, predictions on the CPU, the model back on the CPU after a run unless it is kept
resident."""
import pytest
import torch

import aurora_amd
from aurora_amd import foundry, rollout

# name -> model class, as listed in aurora/foundry/common/model.py:74-150
REFERENCE_REGISTRY = {
    "aurora-0.25-finetuned": "Aurora", "aurora-0.25-pretrained": "AuroraPretrained",
    "aurora-0.25-small-pretrained": "AuroraSmallPretrained", "aurora-0.25-12h-pretrained": "Aurora12hPretrained",
    "aurora-0.1-finetuned": "AuroraHighRes", "aurora-0.4-air-pollution": "AuroraAirPollution", "aurora-0.25-wave": "AuroraWave",
}


def test_registry_lists_the_reference_models_by_name():
    assert set(foundry.models) == set(REFERENCE_REGISTRY)
    for name, cls in foundry.models.items():
        assert issubclass(cls, foundry.Model) and cls.name == name


def test_a_model_without_artifact_fails_with_the_reference_key_error():
    foundry.MLFLOW_ARTIFACTS.clear()
    with pytest.raises(KeyError):
        foundry.models["aurora-0.25-small-pretrained"]()


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    class Tiny(foundry.Model):
        name = "tiny"

        def create_model(self):
            with torch.device("meta"):
                return aurora_amd.AuroraSmallPretrained()

    with pytest.raises(RuntimeError, match="needs a HIP device"):
        Tiny()


@pytest.mark.gpu
@pytest.mark.parametrize("resident", [False, True])
def test_run_yields_the_rollout_on_the_cpu(resident, tmp_path):
    from tests.test_gpu_model import build

    case, model, batch = build("base_pad")
    with torch.inference_mode():
        want = [p.to("cpu") for p in rollout(model, batch, steps=3)]
    ckpt = tmp_path / "model.ckpt"
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, ckpt)
    kwargs = case["kwargs"]

    class Served(foundry.Model):
        name = "served"
        keep_resident = resident

        def create_model(self):
            m = getattr(aurora_amd, case["cls"])(**kwargs)
            m.load_checkpoint_local(foundry.MLFLOW_ARTIFACTS[self.name])
            return m

    foundry.MLFLOW_ARTIFACTS["served"] = str(ckpt)
    served = Served()
    assert next(served.model.parameters()).device.type == "cpu"
    engines = []
    for _ in range(2):   # a second run re-uses (resident) or re-packs (default) the weights: same predictions
        got = list(served.run(batch.to("cpu"), 3))
        engines.append(served.model._engine)
        assert len(got) == 3
        for g, w in zip(got, want):
            assert g.metadata.rollout_step == w.metadata.rollout_step and g.metadata.time == w.metadata.time
            for k, v in w.surf_vars.items():
                assert not g.surf_vars[k].is_cuda and torch.equal(g.surf_vars[k], v), k
            for k, v in w.atmos_vars.items():
                assert torch.equal(g.atmos_vars[k], v), k
        assert next(served.model.parameters()).device.type == ("cuda" if resident else "cpu")
    # resident: ONE handle (packed weights, workspace) served both requests; default: the move to the CPU dropped it
    assert (engines[0] is engines[1] and engines[0] is not None) if resident else engines[1] is None
