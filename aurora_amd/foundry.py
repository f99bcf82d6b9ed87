"""The caller of the path in the reference's serving stack: `Model.run` (aurora/foundry/common/model.py:21-71) and its
registry of named models (`:74-150`), over this package's roll-out.

Same surface -- `models[name]()` creates and loads a model from `MLFLOW_ARTIFACTS[name]`, `run(batch, num_steps)` yields
the predictions on the CPU -- with the two things the reference's loop pays per step moved off the step's critical path:
predictions arrive in pinned host memory through a side stream while the next step computes
(`rollout(..., to_host=True)`: the reference's `pred.to("cpu")` is a synchronous 286 MB copy per step at 0.25 degree), and
with `keep_resident=True` the packed weights stay on the device between runs instead of being moved out and re-packed
(the reference moves the model device <-> CPU around every run, `:62-71`; that remains the default here).
The HTTP / MLflow / Azure plumbing around it (aurora/foundry/server, .../client) is not part of this package.
"""
from __future__ import annotations

import abc
import logging
from typing import Generator

import torch

import aurora_amd
from aurora_amd.batch import Batch
from aurora_amd.rollout import rollout

__all__ = ["Model", "models", "MLFLOW_ARTIFACTS"]

logger = logging.getLogger(__name__)

# `<name, artifact_path>`: absolute path of the checkpoint of every model that may be created (foundry/common/model.py:16-18)
MLFLOW_ARTIFACTS: dict[str, str] = dict()


class Model(metaclass=abc.ABCMeta):
    """A model that can run predictions."""

    keep_resident = False   # True: leave the model (and its packed weights) on the device between runs

    def __init__(self) -> None:
        self.model = self.create_model()
        self.model.eval()
        if not torch.cuda.is_available():
            # The reference falls back to the CPU here; this package has no CPU path (the product fails loudly instead).
            raise RuntimeError("aurora_amd.foundry.Model needs a HIP device: the forward pass runs in the HIP library only")
        self.target_device = torch.device("cuda")

    @abc.abstractmethod
    def create_model(self) -> aurora_amd.Aurora:
        """Create (and load) the model."""

    @torch.inference_mode()
    def run(self, batch: Batch, num_steps: int) -> Generator[Batch, None, None]:
        """Perform `num_steps` prediction steps on the device; the predictions are yielded on the CPU."""
        if not _same_device(next(self.model.parameters()).device, self.target_device):
            self.model.to(self.target_device)   # in place, as upstream (a resident model stays where it is, engine and all)
        batch = batch.to(self.target_device)
        try:
            yield from rollout(self.model, batch, steps=num_steps, to_host=True)
        finally:
            if not self.keep_resident:
                self.model.cpu()


def _same_device(a: torch.device, b: torch.device) -> bool:
    """Type AND index (an index of None means the current device): a model on cuda:1 is not on the target cuda:0."""
    if a.type != b.type:
        return False
    if a.type != "cuda":
        return True
    cur = torch.cuda.current_device()
    return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)


def _named(name: str, cls_name: str) -> type[Model]:
    def create_model(self) -> aurora_amd.Aurora:
        model = getattr(aurora_amd, cls_name)()
        model.load_checkpoint_local(MLFLOW_ARTIFACTS[self.name])
        return model

    return type(cls_name if cls_name != "Aurora" else "AuroraFineTuned", (Model,), {"name": name, "create_model": create_model})


# the registry of the reference, name for name (foundry/common/model.py:74-150)
models: dict[str, type[Model]] = {
    name: _named(name, cls_name)
    for name, cls_name in (
        ("aurora-0.25-finetuned", "Aurora"),
        ("aurora-0.25-pretrained", "AuroraPretrained"),
        ("aurora-0.25-small-pretrained", "AuroraSmallPretrained"),
        ("aurora-0.25-12h-pretrained", "Aurora12hPretrained"),
        ("aurora-0.1-finetuned", "AuroraHighRes"),
        ("aurora-0.4-air-pollution", "AuroraAirPollution"),
        ("aurora-0.25-wave", "AuroraWave"),
    )
}
