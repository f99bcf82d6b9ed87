"""Autoregressive roll-out (reference: aurora/rollout.py:14-49)."""

from __future__ import annotations

import dataclasses
from typing import Generator

import torch

from aurora_amd.batch import Batch

__all__ = ["rollout"]


def rollout(model, batch: Batch, steps: int, graph: bool = False) -> Generator[Batch, None, None]:
    """Yield `steps` successive predictions, feeding each one back as the newest history state.

    The batch is brought to the model's dtype/device once; every prediction stays on the
    device (callers typically `.to("cpu")` what they keep, docs/usage.md:136-141 upstream).

    `graph=True` (not in the reference) captures the step as a hipGraph after a warm-up step and
    replays it: one host call per step instead of ~750 kernel launches; the history is shifted inside
    the graph.  A new graph is captured whenever the LoRA weight set / clamping phase of the step changes.
    """
    if graph:
        yield from _rollout_graphed(model, batch, steps)
        return
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


def _rollout_graphed(model, batch: Batch, steps: int) -> Generator[Batch, None, None]:
    engine = model.engine()
    stepper = None
    state = batch
    for _ in range(steps):
        if stepper is None or engine.step_signature(stepper.state.metadata.rollout_step) != stepper.signature:
            if stepper is not None:
                state = stepper.state   # continue from the captured state with a new weight set
            stepper = engine.capture(state)
        pred = stepper.advance()
        yield pred
        if tuple(pred.surf_vars) != tuple(stepper.state.surf_vars):
            # The ocean-wave variant returns the directions after the other variables, and the eager
            # roll-out (like the reference's) carries on in the prediction's order, which fixes the
            # summation order of the patch embedding.  Follow it: reorder and capture anew.
            st = stepper.state
            state = dataclasses.replace(st, surf_vars={k: st.surf_vars[k] for k in pred.surf_vars})
            stepper = None
