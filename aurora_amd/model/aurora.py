"""The `Aurora` model family, host side.

Mirrors the public surface of the reference's `aurora/model/aurora.py`: the same class
names, constructor keywords and defaults (:55-95, :550-932), `forward(batch) -> Batch`
(:265-392), `load_checkpoint*` (:409-456), the checkpoint-adaptation hooks and the
`state_dict` key schema.  The modules are parameter containers (see schema.py); the step
itself runs in the HIP engine (aurora_amd/engine) and requires the parameters to live on
a HIP device.  There is no CPU implementation in this package: calling `forward` with CPU
parameters raises.
"""

from __future__ import annotations

import dataclasses
import warnings
from datetime import timedelta
from typing import Optional

import torch
from torch import nn

from aurora_amd.batch import Batch
from aurora_amd.model import compat
from aurora_amd.model.schema import AuroraConfig, ParamSpec, ParamTree, param_specs

__all__ = [
    "Aurora",
    "AuroraPretrained",
    "AuroraSmallPretrained",
    "AuroraSmall",
    "Aurora12hPretrained",
    "AuroraHighRes",
    "AuroraAirPollution",
    "AuroraWave",
]


class Aurora(nn.Module):
    """The Aurora model.  Defaults to the 1.3 B parameter configuration."""

    default_checkpoint_repo = "microsoft/aurora"
    default_checkpoint_name = "aurora-0.25-finetuned.ckpt"
    default_checkpoint_revision = "0be7e57c685dac86b78c4a19a3ab149d13c6a3dd"

    #: Which pre/post hooks the engine applies ("base", "air_pollution", "wave").
    variant = "base"

    def __init__(
        self,
        *,
        surf_vars: tuple[str, ...] = ("2t", "10u", "10v", "msl"),
        static_vars: tuple[str, ...] = ("lsm", "z", "slt"),
        atmos_vars: tuple[str, ...] = ("z", "u", "v", "t", "q"),
        window_size: tuple[int, int, int] = (2, 6, 12),
        encoder_depths: tuple[int, ...] = (6, 10, 8),
        encoder_num_heads: tuple[int, ...] = (8, 16, 32),
        decoder_depths: tuple[int, ...] = (8, 10, 6),
        decoder_num_heads: tuple[int, ...] = (32, 16, 8),
        latent_levels: int = 4,
        patch_size: int = 4,
        embed_dim: int = 512,
        num_heads: int = 16,
        mlp_ratio: float = 4.0,
        drop_path: float = 0.0,
        drop_rate: float = 0.0,
        enc_depth: int = 1,
        dec_depth: int = 1,
        dec_mlp_ratio: float = 2.0,
        perceiver_ln_eps: float = 1e-5,
        max_history_size: int = 2,
        timestep: timedelta = timedelta(hours=6),
        stabilise_level_agg: bool = False,
        use_lora: bool = True,
        lora_steps: int = 40,
        lora_mode: str = "single",
        surf_stats: Optional[dict[str, tuple[float, float]]] = None,
        autocast: bool = False,
        bf16_mode: bool = False,
        level_condition: Optional[tuple[int | float, ...]] = None,
        dynamic_vars: bool = False,
        atmos_static_vars: bool = False,
        separate_perceiver: tuple[str, ...] = (),
        modulation_heads: tuple[str, ...] = (),
        positive_surf_vars: tuple[str, ...] = (),
        positive_atmos_vars: tuple[str, ...] = (),
        clamp_at_first_step: bool = False,
        simulate_indexing_bug: bool = False,
    ) -> None:
        super().__init__()
        if drop_path or drop_rate:
            raise NotImplementedError(
                "aurora_amd is an inference engine: drop_path / drop_rate must be 0."
            )
        if latent_levels <= 1:
            raise AssertionError("At least two latent levels are required.")
        if max_history_size <= 0:
            raise AssertionError("At least one history step is required.")
        if sum(encoder_depths) != sum(decoder_depths):
            raise AssertionError("Encoder and decoder must have the same total depth.")

        self.config = AuroraConfig(
            surf_vars=tuple(surf_vars),
            static_vars=tuple(static_vars or ()),
            atmos_vars=tuple(atmos_vars),
            window_size=tuple(window_size),
            encoder_depths=tuple(encoder_depths),
            encoder_num_heads=tuple(encoder_num_heads),
            decoder_depths=tuple(decoder_depths),
            decoder_num_heads=tuple(decoder_num_heads),
            latent_levels=latent_levels,
            patch_size=patch_size,
            embed_dim=embed_dim,
            num_heads=num_heads,
            mlp_ratio=mlp_ratio,
            enc_depth=enc_depth,
            dec_depth=dec_depth,
            dec_mlp_ratio=dec_mlp_ratio,
            perceiver_ln_eps=perceiver_ln_eps,
            max_history_size=max_history_size,
            timestep=timestep,
            stabilise_level_agg=stabilise_level_agg,
            use_lora=use_lora,
            lora_steps=lora_steps,
            lora_mode=lora_mode,
            level_condition=tuple(level_condition) if level_condition else None,
            dynamic_vars=dynamic_vars,
            atmos_static_vars=atmos_static_vars,
            separate_perceiver=tuple(separate_perceiver),
            modulation_heads=tuple(modulation_heads),
            positive_surf_vars=tuple(positive_surf_vars),
            positive_atmos_vars=tuple(positive_atmos_vars),
            clamp_at_first_step=clamp_at_first_step,
            simulate_indexing_bug=simulate_indexing_bug,
        )

        # Attributes the reference exposes and user code reads.
        self.surf_vars = self.config.surf_vars
        self.atmos_vars = self.config.atmos_vars
        self.patch_size = patch_size
        self.surf_stats = surf_stats or dict()
        self.max_history_size = max_history_size
        self.timestep = timestep
        self.use_lora = use_lora
        self.positive_surf_vars = self.config.positive_surf_vars
        self.positive_atmos_vars = self.config.positive_atmos_vars
        self.clamp_at_first_step = clamp_at_first_step

        if self.surf_stats:
            warnings.warn(
                "The normalisation statics for the following surface-level variables are "
                f"manually adjusted: {', '.join(sorted(self.surf_stats))}. "
                "Please ensure that this is right!",
                stacklevel=2,
            )
        if bf16_mode and not autocast:
            warnings.warn(
                "`bf16_mode` was removed; it now activates `autocast` (bf16 backbone).",
                stacklevel=2,
            )
            autocast = True
        self.autocast = autocast

        self.encoder = ParamTree()
        self.backbone = ParamTree()
        self.decoder = ParamTree()
        for spec in param_specs(self.config):
            root, _, _ = spec.name.partition(".")
            getattr(self, root).declare(spec, prefix_to_strip=root + ".")
        # Mirrors of reference attributes that rollout()/user code touch.
        self.encoder.latent_levels = latent_levels
        self.encoder.patch_size = patch_size

        self._engine = None  # built lazily on the first forward (aurora_amd.engine.Engine)
        self._shard = None   # see configure_sharding()

    # -- the step ------------------------------------------------------------------------
    def forward(self, batch: Batch, out=None) -> Batch:
        """One forecast step: `batch` (history of T states) -> prediction at +timestep.

        `out` (not in the reference; used by `rollout`): `(surf, atmos)` dictionaries of preallocated float32 device
        tensors, `(B, 1, H, W)` / `(B, 1, C, H, W)` per variable.  Where such a tensor is contiguous the prediction is
        written into it directly; the returned Batch then holds exactly that tensor."""
        return self.engine().step(batch, out=out)

    def engine(self):
        """The HIP engine bound to this model (created and weight-packed on first use)."""
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError(
                "aurora_amd runs on a HIP device only: move the model with `.to('cuda')` "
                f"before calling it (parameters are on '{p.device}'). There is no CPU path."
            )
        if self._engine is not None and self._engine.is_stale():
            # parameters were modified in place (optimizer step, `p.add_`, `load_state_dict` on a sub-module, ...):
            # pack again.  Edits through `.data` or under `torch.inference_mode()` do not bump the version counters
            # and go unnoticed -- set `model._engine = None` after those.
            self._engine = None
        if self._engine is None:
            from aurora_amd.engine import Engine  # deferred: loads the HIP library

            self._engine = Engine(self)
        return self._engine

    def configure_sharding(self, rank: int, world_size: int, group=None, gather_output: bool = True) -> None:
        """Run ONE forecast across `world_size` GPUs (one process per GPU): the latitude rows of the token
        grid are split into bands, shifted-window attention exchanges halo rows with the neighbouring ranks
        over RCCL point-to-point (aurora_amd/engine/partition.py).  Not part of the reference, which is
        single-device.  Every rank calls `forward` with the same full `Batch`; with
        `gather_output=True` every rank gets the full prediction back, otherwise a `BandBatch` holding its
        own latitude band (which `forward` / `rollout` accept as the next input).
        """
        from aurora_amd.engine.engine import Shard

        self._shard = Shard(rank, world_size, group, gather_output) if world_size > 1 else None
        self._engine = None

    def _apply(self, fn, *args, **kwargs):
        # .to() / .double() / .cuda() that really change storage drop the packed weights; a no-op `.to(device)` on a model
        # that is already there (foundry's `Model.run` does one per request) keeps the handle, its weights and workspace
        # (buffers and in-place writes count too: `_version` moves when `fn` writes a tensor in place)
        def version(t):
            try:
                return t._version
            except RuntimeError:   # inference tensors (a model built under `torch.inference_mode()`) carry no counter
                return -1

        stamp = lambda: [(t.data_ptr(), t.dtype, t.device, version(t))  # noqa: E731
                         for t in (*self.parameters(), *self.buffers())]
        before = stamp()
        out = super()._apply(fn, *args, **kwargs)
        if stamp() != before:
            self._engine = None
        return out

    def load_state_dict(self, *args, **kwargs):
        self._engine = None
        return super().load_state_dict(*args, **kwargs)

    # -- hooks (identity for the base model; subclasses set `variant`) --------------------
    def batch_transform_hook(self, batch: Batch) -> Batch:
        """Transform the batch right after receiving it and before normalisation."""
        return batch

    # -- checkpoints ---------------------------------------------------------------------
    def load_checkpoint(
        self,
        repo: Optional[str] = None,
        name: Optional[str] = None,
        revision: Optional[str] = None,
        strict: bool = True,
    ) -> None:
        """Download a checkpoint from HuggingFace and load it."""
        from huggingface_hub import hf_hub_download

        path = hf_hub_download(
            repo_id=repo or self.default_checkpoint_repo,
            filename=name or self.default_checkpoint_name,
            revision=revision or self.default_checkpoint_revision,
        )
        self.load_checkpoint_local(path, strict=strict)

    def load_checkpoint_local(self, path: str, strict: bool = True) -> None:
        """Load a checkpoint file (published `.ckpt` layouts are adapted on the fly)."""
        device = next(self.parameters()).device
        d = torch.load(path, map_location=device, weights_only=True)
        d = self._adapt_checkpoint(d)

        ckpt_history = d["encoder.surf_token_embeds.weights.2t"].shape[2]
        if self.max_history_size > ckpt_history:
            self.adapt_checkpoint_max_history_size(d)
        elif self.max_history_size < ckpt_history:
            raise AssertionError(
                f"Cannot load checkpoint with `max_history_size` {ckpt_history} "
                f"into model with `max_history_size` {self.max_history_size}."
            )
        self.load_state_dict(d, strict=strict)

    def save_packed(self, path: str) -> None:
        """Write the weights as a packed `AURORAHIP1` file for the C-ABI handle (`aurora_hip_load_packed`): no pickle,
        the big backbone matrices in bf16 when `autocast` is on.  Not in the reference; the model must be on a HIP
        device."""
        self.engine().native.save_packed(path)

    def _adapt_checkpoint(self, d: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        return compat.adapt_pretrained(self.patch_size, d)

    def adapt_checkpoint_max_history_size(self, checkpoint: dict[str, torch.Tensor]) -> None:
        """Zero-extend the history axis of the patch-embedding weights, in place."""
        compat.extend_history(checkpoint, self.max_history_size)

    def configure_activation_checkpointing(self, *args, **kwargs) -> None:
        raise RuntimeError(
            "aurora_amd is a forward/rollout engine: activation checkpointing (a training "
            "feature of the reference) has nothing to checkpoint here."
        )


class AuroraPretrained(Aurora):
    """Pretrained version of Aurora (no LoRA)."""

    default_checkpoint_name = "aurora-0.25-pretrained.ckpt"
    default_checkpoint_revision = "0be7e57c685dac86b78c4a19a3ab149d13c6a3dd"

    def __init__(self, *, use_lora: bool = False, **kw) -> None:
        super().__init__(use_lora=use_lora, **kw)


class AuroraSmallPretrained(Aurora):
    """Small pretrained version of Aurora, for debugging."""

    default_checkpoint_name = "aurora-0.25-small-pretrained.ckpt"
    default_checkpoint_revision = "0be7e57c685dac86b78c4a19a3ab149d13c6a3dd"

    def __init__(
        self,
        *,
        encoder_depths: tuple[int, ...] = (2, 6, 2),
        encoder_num_heads: tuple[int, ...] = (4, 8, 16),
        decoder_depths: tuple[int, ...] = (2, 6, 2),
        decoder_num_heads: tuple[int, ...] = (16, 8, 4),
        embed_dim: int = 256,
        num_heads: int = 8,
        use_lora: bool = False,
        **kw,
    ) -> None:
        super().__init__(
            encoder_depths=encoder_depths,
            encoder_num_heads=encoder_num_heads,
            decoder_depths=decoder_depths,
            decoder_num_heads=decoder_num_heads,
            embed_dim=embed_dim,
            num_heads=num_heads,
            use_lora=use_lora,
            **kw,
        )


AuroraSmall = AuroraSmallPretrained  #: Alias kept for backwards compatibility.


class Aurora12hPretrained(Aurora):
    """Pretrained version of Aurora with a 12-hour time step."""

    default_checkpoint_name = "aurora-0.25-12h-pretrained.ckpt"
    default_checkpoint_revision = "15e76e47b65bf4b28fd2246b7b5b951d6e2443b9"

    def __init__(self, *, timestep: timedelta = timedelta(hours=12), use_lora: bool = False,
                 **kw) -> None:
        super().__init__(timestep=timestep, use_lora=use_lora, **kw)


class AuroraHighRes(Aurora):
    """High-resolution (0.1 degree) version of Aurora: patch size 10."""

    default_checkpoint_name = "aurora-0.1-finetuned.ckpt"
    default_checkpoint_revision = "0be7e57c685dac86b78c4a19a3ab149d13c6a3dd"

    def __init__(
        self,
        *,
        patch_size: int = 10,
        encoder_depths: tuple[int, ...] = (6, 8, 8),
        decoder_depths: tuple[int, ...] = (8, 8, 6),
        **kw,
    ) -> None:
        super().__init__(
            patch_size=patch_size, encoder_depths=encoder_depths, decoder_depths=decoder_depths, **kw
        )


_POLLUTION_SURF = ("pm1", "pm2p5", "pm10", "tcco", "tc_no", "tcno2", "gtco3", "tcso2")
_POLLUTION_ATMOS = ("co", "no", "no2", "go3", "so2")
_CAMS_LEVELS = (50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000)


class AuroraAirPollution(Aurora):
    """Fine-tuned version of Aurora for air pollution (CAMS, 0.4 degree, 12 h step)."""

    default_checkpoint_name = "aurora-0.4-air-pollution.ckpt"
    default_checkpoint_revision = "1764d5630a53d3d7a7d169ca335236fc343e4bfc"
    variant = "air_pollution"

    #: Which history index the predicted difference refers to (reference aurora.py:655-671).
    _predict_difference_history_dim_lookup = {
        "pm1": 0, "pm2p5": 0, "pm10": 0, "co": 1, "tcco": 1, "no": 0, "tc_no": 0,
        "no2": 0, "tcno2": 0, "so2": 1, "tcso2": 1, "go3": 1, "gtco3": 1,
    }

    def __init__(
        self,
        *,
        surf_vars: tuple[str, ...] = ("2t", "10u", "10v", "msl") + _POLLUTION_SURF,
        static_vars: tuple[str, ...] = (
            ("lsm", "z", "slt")
            + ("static_ammonia", "static_ammonia_log", "static_co", "static_co_log")
            + ("static_nox", "static_nox_log", "static_so2", "static_so2_log")
        ),
        atmos_vars: tuple[str, ...] = ("z", "u", "v", "t", "q") + _POLLUTION_ATMOS,
        patch_size: int = 3,
        timestep: timedelta = timedelta(hours=12),
        level_condition: Optional[tuple[int | float, ...]] = _CAMS_LEVELS,
        dynamic_vars: bool = True,
        atmos_static_vars: bool = True,
        separate_perceiver: tuple[str, ...] = _POLLUTION_ATMOS,
        modulation_heads: tuple[str, ...] = tuple(_predict_difference_history_dim_lookup),
        positive_surf_vars: tuple[str, ...] = _POLLUTION_SURF,
        positive_atmos_vars: tuple[str, ...] = _POLLUTION_ATMOS,
        simulate_indexing_bug: bool = True,
        **kw,
    ) -> None:
        super().__init__(
            surf_vars=surf_vars,
            static_vars=static_vars,
            atmos_vars=atmos_vars,
            patch_size=patch_size,
            timestep=timestep,
            level_condition=level_condition,
            dynamic_vars=dynamic_vars,
            atmos_static_vars=atmos_static_vars,
            separate_perceiver=separate_perceiver,
            modulation_heads=modulation_heads,
            positive_surf_vars=positive_surf_vars,
            positive_atmos_vars=positive_atmos_vars,
            simulate_indexing_bug=simulate_indexing_bug,
            **kw,
        )
        # Linear(2, 1) feature combiners of the log transform, weight 0.5 / bias 0.
        self.surf_feature_combiner = ParamTree()
        self.atmos_feature_combiner = ParamTree()
        for tree, names in (
            (self.surf_feature_combiner, self.positive_surf_vars),
            (self.atmos_feature_combiner, self.positive_atmos_vars),
        ):
            for v in names:
                tree.declare(ParamSpec(f"{v}.weight", (1, 2), "half"))
                tree.declare(ParamSpec(f"{v}.bias", (1,), "zeros"))

    def _adapt_checkpoint(self, d):
        return compat.adapt_air_pollution(self.patch_size, compat.adapt_pretrained(self.patch_size, d))


_WAVE_VARS = (
    ("swh", "mwd", "mwp", "pp1d", "shww", "mdww", "mpww", "shts", "mdts", "mpts")
    + ("swh1", "mwd1", "mwp1", "swh2", "mwd2", "mwp2", "wind", "10u_wave", "10v_wave")
)


class AuroraWave(Aurora):
    """Version of Aurora fine-tuned to HRES-WAM ocean wave data."""

    default_checkpoint_name = "aurora-0.25-wave.ckpt"
    default_checkpoint_revision = "74598e8c65d53a96077c08bb91acdfa5525340c9"
    variant = "wave"

    def __init__(
        self,
        *,
        surf_vars: tuple[str, ...] = ("2t", "10u", "10v", "msl") + _WAVE_VARS,
        static_vars: tuple[str, ...] = ("lsm", "z", "slt", "wmb", "lat_mask"),
        lora_mode: str = "from_second",
        stabilise_level_agg: bool = True,
        density_channel_surf_vars: tuple[str, ...] = _WAVE_VARS,
        angle_surf_vars: tuple[str, ...] = ("mwd", "mdww", "mdts", "mwd1", "mwd2"),
        **kw,
    ) -> None:
        # The model sees sin/cos pairs for angles and an extra density channel per wave var.
        expanded: tuple[str, ...] = ()
        for name in surf_vars:
            expanded += (f"{name}_sin", f"{name}_cos") if name in angle_surf_vars else (name,)
            if name in density_channel_surf_vars:
                expanded += (f"{name}_density",)
        super().__init__(
            surf_vars=expanded,
            static_vars=static_vars,
            lora_mode=lora_mode,
            stabilise_level_agg=stabilise_level_agg,
            **kw,
        )
        self.density_channel_surf_vars = tuple(density_channel_surf_vars)
        self.angle_surf_vars = tuple(angle_surf_vars)

    def _adapt_checkpoint(self, d):
        return compat.adapt_wave(self.patch_size, compat.adapt_pretrained(self.patch_size, d))

    def batch_transform_hook(self, batch: Batch) -> Batch:
        from aurora_amd.model import wave

        return wave.transform_batch(batch)
