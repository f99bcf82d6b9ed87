"""Autoregressive roll-out (reference: aurora/rollout.py:14-49)."""

from __future__ import annotations

import dataclasses
from typing import Generator

import torch

from aurora_amd.batch import Batch

__all__ = ["rollout"]


def rollout(model, batch: Batch, steps: int) -> Generator[Batch, None, None]:
    """Yield `steps` successive predictions, feeding each one back as the newest history state.

    The batch is brought to the model's dtype/device once; every prediction stays on the
    device (callers typically `.to("cpu")` what they keep, docs/usage.md:136-141 upstream).
    """
    batch = model.batch_transform_hook(batch)
    p = next(model.parameters())
    batch = batch.type(p.dtype).crop(model.patch_size).to(p.device)
    shard = getattr(model, "_shard", None)
    if shard is not None and not shard.gather_output:
        # sharded model that keeps its state distributed: continue from this rank's latitude band
        batch = model.engine().local_band(batch)

    for _ in range(steps):
        pred = model.forward(batch)
        yield pred
        # Newest state in, oldest state out.  `pred` carries time and rollout_step.
        batch = dataclasses.replace(
            pred,
            surf_vars={
                k: torch.cat([batch.surf_vars[k][:, 1:], v], dim=1)
                for k, v in pred.surf_vars.items()
            },
            atmos_vars={
                k: torch.cat([batch.atmos_vars[k][:, 1:], v], dim=1)
                for k, v in pred.atmos_vars.items()
            },
        )
