"""Autoregressive roll-out (reference: aurora/rollout.py:14-49)."""

from __future__ import annotations

import dataclasses
from collections import deque
from typing import Generator

import torch

from aurora_amd.batch import Batch

__all__ = ["rollout", "write_rollout"]

_RING = 8   # history states per ring chunk


class _History:
    """The history of one variable, (B, T, ...) per step, WITHOUT the reference's per-step `torch.cat`
    (aurora/rollout.py:39-49: `cat([old[:, 1:], pred], dim=1)` re-copies the T-1 surviving states of every variable at
    every step).  States live in a chunk (B, R, ...) along the history axis; the window handed to the model is the view
    chunk[:, i:i+T], a prediction is written into slot i+T and the window slides by one -- no copy.  When a chunk is
    full a fresh one is allocated and seeded with the T-1 newest states (one copy per R-T+1 steps); finished chunks are
    never overwritten, so predictions handed to the caller (views into them) stay valid for as long as they are held."""

    def __init__(self, x: torch.Tensor) -> None:
        self.T = x.shape[1]
        self.chunk = x.new_empty((x.shape[0], max(_RING, self.T + 1), *x.shape[2:]))
        self.chunk[:, :self.T].copy_(x)
        self.i = 0

    def window(self) -> torch.Tensor:
        return self.chunk[:, self.i:self.i + self.T]

    def slot(self) -> torch.Tensor:
        """Where the next prediction (B, 1, ...) goes.  The model writes it there itself when it can (`out=`)."""
        T, R = self.T, self.chunk.shape[1]
        self._fresh = None
        if self.i + T == R:   # chunk full: continue in a new one, seeded with the T-1 states that stay in the window
            self._fresh = torch.empty_like(self.chunk)
            self._fresh[:, :T - 1].copy_(self.chunk[:, self.i + 1:])
            return self._fresh[:, T - 1:T]
        return self.chunk[:, self.i + T:self.i + T + 1]

    def advance(self) -> None:
        if self._fresh is not None:
            self.chunk, self.i, self._fresh = self._fresh, 0, None
        else:
            self.i += 1


def rollout(model, batch: Batch, steps: int, graph: bool = False, to_host: bool = False) -> Generator[Batch, None, None]:
    """Yield `steps` successive predictions, feeding each one back as the newest history state.

    The batch is brought to the model's dtype/device once; every prediction stays on the
    device (callers typically `.to("cpu")` what they keep, docs/usage.md:136-141 upstream).

    Aliasing (differs from the reference, whose `torch.cat` copies): a yielded prediction is a VIEW into a history chunk
    of 8 states that is also the model's next input.  Holding one prediction keeps its whole chunk alive (2.3 GB at 0.25
    degree against 286 MB for the prediction), and editing a yielded tensor in place changes what later steps read.
    Callers who keep predictions on the device should `.clone()` them (or move them: `.to("cpu")`, `to_host=True`).

    `graph=True` (not in the reference) captures the step as a hipGraph after a warm-up step and
    replays it: one host call per step instead of ~750 kernel launches; the history is shifted inside
    the graph.  A new graph is captured whenever the LoRA weight set / clamping phase of the step changes.

    `to_host=True` (not in the reference) yields the predictions in pinned HOST memory instead: the device->host
    copy of step s (286 MB at 0.25 degree) runs on a side stream while step s+1 computes, so the caller's
    `[p.to("cpu") for p in rollout(...)]` pattern costs no step time.  Predictions arrive one step late (the generator
    runs one step ahead); nothing else changes.
    """
    if to_host:
        yield from _to_host(rollout(model, batch, steps, graph=graph))
        return
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

    hist_s = {k: _History(v) for k, v in batch.surf_vars.items()}
    hist_a = {k: _History(v) for k, v in batch.atmos_vars.items()}
    for _ in range(steps):
        # Newest state in, oldest state out -- by sliding a window over the history chunks: the model writes its
        # prediction straight into the next slot (no copy at all) when that slot is a plain contiguous field (B = 1),
        # otherwise the prediction is copied in.  `pred` carries time and rollout_step.
        slots = ({k: h.slot() for k, h in hist_s.items()}, {k: h.slot() for k, h in hist_a.items()})
        pred = model.forward(batch, out=slots)
        surf, atmos = {}, {}
        for hist, src, dst in ((hist_s, pred.surf_vars, surf), (hist_a, pred.atmos_vars, atmos)):
            for k, v in src.items():
                slot = slots[0 if hist is hist_s else 1][k]
                if v.data_ptr() != slot.data_ptr():
                    slot.copy_(v)
                hist[k].advance()
                dst[k] = slot
        pred = dataclasses.replace(pred, surf_vars=surf, atmos_vars=atmos)
        yield pred
        batch = dataclasses.replace(
            pred,
            surf_vars={k: hist_s[k].window() for k in pred.surf_vars},
            atmos_vars={k: hist_a[k].window() for k in pred.atmos_vars},
        )


def _to_host(preds) -> Generator[Batch, None, None]:
    """Device predictions -> pinned host memory, asynchronously, one step behind the computation."""
    side = torch.cuda.Stream()
    pending: deque = deque()

    def finish(item):
        host, done, _keep = item
        done.synchronize()
        return host

    for pred in preds:
        ready = torch.cuda.Event()
        ready.record()                       # the step's kernels are enqueued on the current stream
        with torch.cuda.stream(side):
            side.wait_event(ready)
            pin = lambda v: torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v, non_blocking=True)  # noqa: E731
            host = dataclasses.replace(
                pred,
                surf_vars={k: pin(v) for k, v in pred.surf_vars.items()},
                static_vars={k: v for k, v in pred.static_vars.items()},
                atmos_vars={k: pin(v) for k, v in pred.atmos_vars.items()},
            )
            done = torch.cuda.Event()
            done.record(side)
        pending.append((host, done, pred))   # `pred` stays referenced until its copy has finished
        if len(pending) > 1:
            yield finish(pending.popleft())
    while pending:
        yield finish(pending.popleft())


def _rollout_graphed(model, batch: Batch, steps: int) -> Generator[Batch, None, None]:
    engine = model.engine()
    stepper = None
    state = batch
    for _ in range(steps):
        if (stepper is None or engine.step_signature(stepper.state.metadata.rollout_step) != stepper.signature
                or stepper._handle_addresses() != stepper._addresses):   # (the handle re-allocated what the graph points at)
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


def _fill(template: str, **known) -> str:
    """`template.format(**known)` that leaves the fields it does not know (e.g. `{rank:02d}`) as they are."""
    import string

    out = []
    for text, field, spec, conv in string.Formatter().parse(template):
        out.append(text)
        if field is None:
            continue
        if field in known:
            out.append(format(known[field], spec or ""))
        else:
            out.append("{" + field + ("!" + conv if conv else "") + (":" + spec if spec else "") + "}")
    return "".join(out)


def write_rollout(model, batch: Batch, steps: int, path_template: str, graph: bool = False) -> list[str]:
    """Roll out `steps` predictions and write each to `path_template` (fields `{step}`, and `{rank}` for a sharded model
    that keeps its state distributed: every rank writes its own latitude band, nothing is gathered -- see
    `BandBatch.to_netcdf`).  The output path of SURVEY.md section 8 f-4, built on `rollout(..., to_host=True)`: the
    device -> pinned-host copy of step s runs on a side stream under step s+1, and the file of step s is written by a
    background thread meanwhile, so neither transfer nor encoding holds the next step up.  Returns the paths written
    by this process.  (The reference's pattern is `[p.to("cpu") for p in rollout(...)]` + `Batch.to_netcdf`,
    docs/usage.md:136-141, aurora/batch.py:224-257.)"""
    import queue
    import threading

    if "{step" not in path_template:
        raise ValueError("the path template needs a `{step}` field")
    todo: "queue.Queue" = queue.Queue(maxsize=2)
    written, errors = [], []

    def writer():
        while True:
            item = todo.get()
            if item is None:
                return
            pred, path = item
            try:
                if not hasattr(pred, "rank"):   # a plain Batch (un-sharded, or gathered): `{rank}` is rank 0, filled HERE --
                    path = _fill(path, rank=0)  # Batch.to_netcdf would take the braces literally
                pred.to_netcdf(path)
                written.append(_fill(path, rank=getattr(pred, "rank", 0)))
            except Exception as e:  # noqa: BLE001  (re-raised in the caller's thread below)
                errors.append(e)

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    try:
        for pred in rollout(model, batch, steps, graph=graph, to_host=True):
            if errors:
                break
            todo.put((pred, _fill(path_template, step=pred.metadata.rollout_step)))
    finally:
        todo.put(None)
        th.join()
    if errors:
        raise errors[0]
    return written
