"""The MI355X engine behind `Aurora.forward`: a thin binding of the C-ABI model handle.

`Engine.step(batch)` is what `Aurora.forward` executes.  The data flow of the reference's forward
(aurora/model/aurora.py:265-392; encoder.py:198-366; swin3d.py:884-936, 440-509; decoder.py:168-276) -- for every
model class, on one device or on one latitude band of a forecast sharded over several -- is sequenced by the handle
itself (aurora_amd/csrc/step.hip, include/aurora_hip.h): ONE sequencer, in C++, that any host language can drive.  What
is left here is Python-side plumbing: `Batch` <-> raw pointers (engine/native.py), the halo transport of sharded steps
(torch.distributed point-to-point on RCCL), gathering a sharded prediction, per-launch timing for bench.py, and
hipGraph capture of a step.

Precision: parameters must be fp32.  With `autocast=False` everything runs in fp32 (large fp32 linears by exact operand
splitting on the MFMA pipe).  With `autocast=True` the backbone GEMMs and attention take bf16 operands with fp32
accumulation while LayerNorm statistics, softmax and the residual stream stay fp32 -- the semantics of the reference's
`torch.autocast` region (aurora.py:327-343); encoder and decoder stay fp32 like upstream.

There is no CPU or torch fallback: `lib.load()` raises if libaurora_hip.so is missing.
"""

from __future__ import annotations

import ctypes
import dataclasses
from typing import Optional

import numpy as np
import torch

from aurora_amd.batch import BandBatch, Batch, derive_metadata
from aurora_amd.engine import lib, native

F32 = torch.float32


@dataclasses.dataclass
class Shard:
    """Latitude-band sharding of one forecast over `world` ranks (csrc/band.hip; numpy twin: engine/partition.py)."""

    rank: int
    world: int
    group: object = None          # torch.distributed process group (None: default group)
    gather_output: bool = True    # True: forward() returns the full fields on every rank


class Engine:
    def __init__(self, model, transport=None) -> None:
        lib.load()  # fail loudly if libaurora_hip.so is missing
        self.model = model
        self.cfg = cfg = model.config
        p = next(model.parameters())
        if p.dtype not in (F32, torch.float64):   # float64 masters are rounded to float32 with a warning (native.py)
            raise NotImplementedError(
                f"aurora_amd computes in fp32 (or bf16 backbone with autocast=True); parameters "
                f"are {p.dtype}. Keep the model in float32 (or float64)."
            )
        self.device = p.device
        self._rows_cache: dict = {}
        for heads, dim in zip(cfg.encoder_num_heads + cfg.decoder_num_heads,
                              cfg.stage_dims() + cfg.stage_dims()[::-1]):
            if dim != heads * 64:
                raise NotImplementedError(
                    f"the window-attention kernel is built for head_dim 64 (dim {dim}, {heads} heads)"
                )
        if int(np.prod(cfg.window_size)) > 144:
            raise NotImplementedError("windows of more than 144 tokens are not supported")
        self._param_stamp = self._stamp()
        shard = getattr(model, "_shard", None)
        self.shard: Optional[Shard] = shard if (shard is not None and shard.world > 1) else None
        self.native = native.NativeModel(model, self.shard, transport)
        self._capturing = False        # True while a hipGraph of the step is being captured

    # ---------------------------------------------------------------------------------------
    def _stamp(self) -> int:
        try:
            return sum(p._version for p in self.model.parameters())
        except RuntimeError:  # inference tensors carry no version counter
            return -1

    def is_stale(self) -> bool:
        return self._stamp() != self._param_stamp

    def _lora_key(self, step: int):
        """Which merged weight set roll-out step `step` uses (reference lora.py:105-129)."""
        cfg = self.cfg
        if not cfg.use_lora or step >= cfg.lora_steps:
            return "base"
        if cfg.lora_mode == "single":
            return 0
        if cfg.lora_mode == "from_second":
            return "base" if step == 0 else 0
        if cfg.lora_mode == "all":
            return step
        raise ValueError(f"Invalid mode: {cfg.lora_mode}")

    def step_signature(self, step: int):
        """Everything about a step that depends on the roll-out step and is baked into a captured graph:
        the LoRA weight set and whether positive variables are clamped (aurora.py:368-388)."""
        cfg = self.cfg
        new_step = step + 1
        clamp = bool(cfg.positive_surf_vars or cfg.positive_atmos_vars) and (
            new_step >= 1 if cfg.clamp_at_first_step else new_step > 1)
        return (self._lora_key(step), clamp)

    # ---------------------------------------------------------------------------------------
    # the step
    # ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, batch: Batch, out=None) -> Batch:
        """One forecast step.  With sharding: on this rank's latitude band, halo rows exchanged by RCCL point-to-point
        from inside the handle.  `out`: optional destination tensors of the prediction, see `Aurora.forward`."""
        model, cfg = self.model, self.cfg
        if not self._capturing:  # (a captured step runs on a batch the hook has already seen)
            batch = model.batch_transform_hook(batch)
        if self.shard is not None:
            # the caller's own full-grid static fields never change: a gathered forecast hands them back as they came
            full_static = None
            if self.shard.gather_output and not isinstance(batch, BandBatch):
                batch = batch.type(F32).crop(cfg.patch_size).to(self.device)   # (local_band's own conversions are no-ops then)
                full_static = dict(batch.static_vars)
            band = self.local_band(batch)
            T = next(iter(band.surf_vars.values())).shape[1]
            assert T <= cfg.max_history_size, f"{T} > {cfg.max_history_size}."
            pred = self.native.step(band, upload_time=True, out=None if self.shard.gather_output else out)
            return self._gather(pred, full_static) if self.shard.gather_output else pred
        batch = batch.type(F32).crop(cfg.patch_size).to(self.device)
        T = next(iter(batch.surf_vars.values())).shape[1]
        assert T <= cfg.max_history_size, f"{T} > {cfg.max_history_size}."
        return self.native.step(batch, upload_time=not self._capturing, out=out)

    def local_band(self, batch: Batch) -> BandBatch:
        """This rank's latitude band of a batch, as views (float32, on the device).  A full batch also tells the handle
        the grid (`aurora_hip_precompute` with every rank's latitudes: the partition is the handle's)."""
        cfg, sh = self.cfg, self.shard
        assert sh is not None, "local_band() is for models with configure_sharding()"
        if isinstance(batch, BandBatch):
            assert self.native.band_rows is not None, "a latitude band was given before the grid itself"
            return batch.type(F32).to(self.device)
        batch = batch.type(F32).crop(cfg.patch_size).to(self.device)
        md = batch.metadata
        assert md.lat.dim() == 1, "latitude bands need vector coordinates"
        self.native.precompute(md.lat, md.lon, tuple(md.atmos_levels))
        r0, r1 = self.native.band_rows
        P = cfg.patch_size
        cut = lambda d_: {k: v[..., r0:r1, :] for k, v in d_.items()}  # noqa: E731
        return BandBatch(cut(batch.surf_vars), cut(batch.static_vars), cut(batch.atmos_vars),
                         derive_metadata(md, lat=md.lat[r0:r1]), full_patch_rows=batch.spatial_shape[0] // P,
                         band=(r0 // P, r1 // P), rank=sh.rank, world=sh.world)

    def band_rows_of(self, rank: int, full_patch_rows: int, patch_cols: int) -> tuple[int, int]:
        """Patch rows [h0, h1) of `rank`'s band (the handle's own partition function; pure host code).  Cached per grid:
        the partition is a search, and a gathered forecast asks for every rank's rows every step."""
        key = (rank, full_patch_rows, patch_cols)
        hit = self._rows_cache.get(key)
        if hit is not None:
            return hit
        self._rows_cache[key] = out = self._band_rows_of(rank, full_patch_rows, patch_cols)
        return out

    def _band_rows_of(self, rank: int, full_patch_rows: int, patch_cols: int) -> tuple[int, int]:
        cfg = self.cfg
        i32 = lambda v: (ctypes.c_int32 * len(v))(*v)  # noqa: E731
        h0, h1 = ctypes.c_int32(), ctypes.c_int32()
        lib._check(lib.load().aurora_hip_band_partition(
            len(cfg.encoder_depths), i32((cfg.latent_levels, full_patch_rows, patch_cols)), i32(tuple(cfg.window_size)),
            self.shard.world, rank, 0, ctypes.byref(h0), ctypes.byref(h1)))
        return h0.value, h1.value

    def _gather(self, pred: BandBatch, full_static=None) -> Batch:
        """Assemble the full fields on every rank with ONE collective: every field of the band is packed into one buffer
        (bands padded to the tallest), all-gathered, and cut back into place.  `full_static`: the full-grid static fields
        the caller passed in -- they never change, so they are handed back instead of being packed and gathered."""
        import torch.distributed as dist

        sh, P = self.shard, self.cfg.patch_size
        W = pred.spatial_shape[1]
        rows0 = [self.band_rows_of(r, pred.full_patch_rows, W // P) for r in range(sh.world)]
        heights = [(b - a) * P for a, b in rows0]
        h, h_max, H = heights[sh.rank], max(heights), rows0[-1][1] * P
        groups = (pred.surf_vars, {} if full_static is not None else pred.static_vars, pred.atmos_vars)
        items = [(gi, k, v) for gi, d_ in enumerate(groups) for k, v in d_.items()]
        planes = [v.numel() // (h * W) for _, _, v in items]
        mine = torch.zeros((sum(planes), h_max, W), dtype=F32, device=self.device)
        at = 0
        for (_, _, v), n in zip(items, planes):
            mine[at:at + n, :h] = v.reshape(n, h, W)
            at += n
        if dist.get_backend(sh.group) == "gloo":   # tests: staged through host memory
            parts = [torch.empty(mine.shape, dtype=F32) for _ in range(sh.world)]
            dist.all_gather(parts, mine.cpu(), group=sh.group)
            everyone = torch.stack(parts).to(self.device)
        else:
            everyone = torch.empty((sh.world, *mine.shape), dtype=F32, device=self.device)
            dist.all_gather_into_tensor(everyone, mine, group=sh.group)
        out: tuple[dict, dict, dict] = ({}, {}, {})
        at = 0
        for (gi, k, v), n in zip(items, planes):
            full = torch.cat([everyone[r, at:at + n, :heights[r]] for r in range(sh.world)], dim=1)
            out[gi][k] = full.reshape(*v.shape[:-2], H, W)
            at += n
        lat = self.native.full_lat()   # the whole (cropped) grid went through aurora_hip_precompute on every rank
        assert lat is not None and lat.shape[0] == H, "the full grid's latitudes are unknown to this rank"
        # (the caller's own full-grid static fields were not gathered: they are handed back as they came -- the next roll-out
        # step, `_to_host` and `write_rollout` all read them from the prediction)
        static = dict(full_static) if full_static is not None else out[1]
        return Batch(out[0], static, out[2], derive_metadata(pred.metadata, lat=lat.to(pred.metadata.lat)))

    # -- per-launch timing (bench.py, tools): HIP events on the launch stream, inside the handle ---------------------
    def profile_start(self, only=None) -> None:
        """Bracket every launch (or the kernels named in `only`) with HIP events until `profile_stop`."""
        self.native.profile_begin(only)

    def profile_stop(self) -> dict:
        """{kernel: {"launches", "ms", "work"}} of the launches since `profile_start`."""
        return self.native.profile_end()

    def capture(self, batch: Batch) -> "GraphedStep":
        """Capture one step on (a private copy of) `batch` into a hipGraph; see GraphedStep."""
        return GraphedStep(self, batch)


class GraphedStep:
    """One forecast step captured as a hipGraph (BASELINE config 3: roll-out with a captured step).

    The ~750 kernel launches of a step are recorded once (`torch.cuda.CUDAGraph`; the handle launches on torch's current
    stream, which is the capture stream) and replayed with one host call.  The graph reads its inputs from private
    static buffers and, as its last nodes, shifts the history in place (oldest state out, prediction in), so consecutive
    `advance()` calls ARE the roll-out.  Only the clock-dependent inputs change from step to step; they are uploaded into
    the handle's persistent buffers before each replay (`aurora_hip_set_time_ex`, outside the graph).  The LoRA weight set
    and the positive-variable clamp are baked in: the roll-out re-captures when `Engine.step_signature` changes
    (lora.py:105-129: after `lora_steps`, or after the first step in "from_second" mode; aurora.py:368-388: clamping
    starts at the second step).  The graph also bakes in the addresses of the handle's workspace and tables: `advance`
    refuses to replay after the handle has re-allocated them (a larger batch, another grid) -- capture anew then.
    """

    def __init__(self, engine: Engine, batch: Batch) -> None:
        assert engine.shard is None, "graph capture of sharded steps is not supported (RCCL point-to-point inside a capture)"
        self.engine = engine
        cfg = engine.cfg
        batch = engine.model.batch_transform_hook(batch)
        batch = batch.type(F32).crop(cfg.patch_size).to(engine.device)
        clone = lambda d_: {k: v.clone() for k, v in d_.items()}  # noqa: E731
        self.state = Batch(clone(batch.surf_vars), clone(batch.static_vars), clone(batch.atmos_vars),
                           derive_metadata(batch.metadata, lat=batch.metadata.lat.clone(),
                                           lon=batch.metadata.lon.clone()))
        self.signature = engine.step_signature(self.state.metadata.rollout_step)
        # Warm every cache (tables, grids, weight sets, the workspace) with an eager step, then capture.
        engine.step(self.state)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        engine.native.set_time(self.state.metadata.time)
        engine._capturing = True
        try:
            with torch.cuda.graph(self.graph):
                self.pred = engine.step(self.state)
                # History shift, inside the graph: the captured step reads fixed addresses, so the states move, not the
                # window (slot t <- slot t+1 in ascending order: no temporary; then the prediction into the last slot).
                for preds, states in ((self.pred.surf_vars, self.state.surf_vars),
                                      (self.pred.atmos_vars, self.state.atmos_vars)):
                    for k, v in preds.items():
                        x = states[k]
                        for t in range(x.shape[1] - 1):
                            x[:, t].copy_(x[:, t + 1])
                        x[:, -1:].copy_(v)
        finally:
            engine._capturing = False
        self._addresses = self._handle_addresses()

    def _handle_addresses(self):
        return self.engine.native.generation()

    def advance(self) -> Batch:
        """Replay the graph once: returns the prediction (fresh tensors) and moves the state forward."""
        eng, md = self.engine, self.state.metadata
        assert eng.step_signature(md.rollout_step) == self.signature, "roll-out phase changed: capture a new graph"
        if self._handle_addresses() != self._addresses:
            raise RuntimeError("the model handle re-allocated its workspace or grid tables after this graph was captured "
                               "(a larger batch or another grid ran on the same model): capture a new graph")
        eng.native.set_time(md.time)          # the only step-dependent inputs
        self.graph.replay()
        new_md = derive_metadata(self.pred.metadata, time=tuple(t + eng.cfg.timestep for t in md.time),
                                 rollout_step=md.rollout_step + 1)
        out = Batch({k: v.clone() for k, v in self.pred.surf_vars.items()}, dict(self.pred.static_vars),
                    {k: v.clone() for k, v in self.pred.atmos_vars.items()}, new_md)
        self.state = dataclasses.replace(self.state, metadata=derive_metadata(
            md, time=new_md.time, rollout_step=new_md.rollout_step))
        return out
