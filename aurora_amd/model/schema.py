"""Declarative parameter schema of the Aurora model family.

The drop-in boundary of this engine is the reference's `state_dict` key schema
(SURVEY.md §8b): published checkpoints must load with `load_state_dict(strict=True)`.
Instead of restating the reference's nested module classes, the whole tree is described
here as a flat list of `(dotted_name, shape, init)` records derived from the constructor
arguments (reference: aurora/model/aurora.py:55-263, encoder.py:33-170,
swin3d.py:75-134,363-438,512-613,751-866, decoder.py:29-138, perceiver.py:91-210,
patchembed.py:18-77, film.py:17-36, lora.py:17-103, levelcond.py:16-35).

`ParamTree` materialises such a list as nested `nn.Module` containers whose
`state_dict()` keys are exactly the dotted names.  The modules hold parameters only;
the compute lives in the HIP engine (aurora_amd/engine).
"""

from __future__ import annotations

import dataclasses
import math
from datetime import timedelta
from typing import Iterator, Optional

import torch
from torch import nn

from aurora_amd.normalisation import level_to_str

__all__ = ["AuroraConfig", "ParamSpec", "param_specs", "ParamTree", "DYNAMIC_VARS"]

DYNAMIC_VARS = ("tod_cos", "tod_sin", "dow_cos", "dow_sin", "doy_cos", "doy_sin")
LORA_RANK = 8
LORA_ALPHA = 8


@dataclasses.dataclass(frozen=True)
class AuroraConfig:
    """Constructor arguments of `Aurora` (reference aurora.py:55-95), frozen."""

    surf_vars: tuple[str, ...]
    static_vars: tuple[str, ...]
    atmos_vars: tuple[str, ...]
    window_size: tuple[int, int, int]
    encoder_depths: tuple[int, ...]
    encoder_num_heads: tuple[int, ...]
    decoder_depths: tuple[int, ...]
    decoder_num_heads: tuple[int, ...]
    latent_levels: int
    patch_size: int
    embed_dim: int
    num_heads: int
    mlp_ratio: float
    enc_depth: int
    dec_depth: int
    dec_mlp_ratio: float
    perceiver_ln_eps: float
    max_history_size: int
    timestep: timedelta
    stabilise_level_agg: bool
    use_lora: bool
    lora_steps: int
    lora_mode: str
    level_condition: Optional[tuple[int | float, ...]]
    dynamic_vars: bool
    atmos_static_vars: bool
    separate_perceiver: tuple[str, ...]
    modulation_heads: tuple[str, ...]
    positive_surf_vars: tuple[str, ...]
    positive_atmos_vars: tuple[str, ...]
    clamp_at_first_step: bool
    simulate_indexing_bug: bool

    # -- derived variable lists ----------------------------------------------------------
    @property
    def enc_static_vars(self) -> tuple[str, ...]:
        """Static variables seen by the encoder (dynamic ones appended, encoder.py:103-106)."""
        sv = tuple(self.static_vars or ())
        return sv + DYNAMIC_VARS if self.dynamic_vars else sv

    @property
    def enc_surf_vars(self) -> tuple[str, ...]:
        return tuple(self.surf_vars) + self.enc_static_vars

    @property
    def enc_atmos_vars(self) -> tuple[str, ...]:
        av = tuple(self.atmos_vars)
        if self.enc_static_vars and self.atmos_static_vars:
            av += tuple(f"static_{v}" for v in self.enc_static_vars)
        return av

    @property
    def dec_surf_vars(self) -> tuple[str, ...]:
        sv = tuple(self.surf_vars)
        return sv + tuple(f"{v}_mod" for v in sv if v in self.modulation_heads)

    @property
    def dec_atmos_vars(self) -> tuple[str, ...]:
        av = tuple(self.atmos_vars)
        return av + tuple(f"{v}_mod" for v in av if v in self.modulation_heads)

    @property
    def dec_separate_perceiver(self) -> tuple[str, ...]:
        sp = tuple(self.separate_perceiver)
        if self.modulation_heads:
            sp += tuple(f"{v}_mod" for v in sp)
        return sp

    @property
    def num_lora_sets(self) -> int:
        return self.lora_steps if self.lora_mode == "all" else 1

    def stage_dims(self) -> list[int]:
        return [self.embed_dim * 2**i for i in range(len(self.encoder_depths))]


@dataclasses.dataclass(frozen=True)
class ParamSpec:
    name: str
    shape: tuple[int, ...]
    init: str  # one of the keys of _INITS
    fan_in: int = 0


def _linear(prefix: str, n_out: int, n_in: int, bias: bool = True, init: str = "trunc02"):
    yield ParamSpec(f"{prefix}.weight", (n_out, n_in), init)
    if bias:
        yield ParamSpec(f"{prefix}.bias", (n_out,), "zeros")


def _layernorm(prefix: str, dim: int):
    yield ParamSpec(f"{prefix}.weight", (dim,), "ones")
    yield ParamSpec(f"{prefix}.bias", (dim,), "zeros")


def _patch_embed(prefix: str, var_names, dim: int, t: int, p: int):
    fan_in = t * p * p
    for v in var_names:
        yield ParamSpec(f"{prefix}.weights.{v}", (dim, 1, t, p, p), "kaiming5", fan_in)
    yield ParamSpec(f"{prefix}.bias", (dim,), "uniform_fan", fan_in)


def _resampler(prefix: str, dim: int, depth: int, head_dim: int, heads: int, ratio: float,
               ln_k_q: bool):
    inner = head_dim * heads
    for i in range(depth):
        a = f"{prefix}.layers.{i}"
        yield from _linear(f"{a}.0.to_q", inner, dim, bias=False)
        yield from _linear(f"{a}.0.to_kv", 2 * inner, dim, bias=False)
        yield from _linear(f"{a}.0.to_out", dim, inner, bias=False)
        if ln_k_q and i == 0:
            yield from _layernorm(f"{a}.0.ln_k", inner)
            yield from _layernorm(f"{a}.0.ln_q", inner)
        hidden = int(dim * ratio)
        yield from _linear(f"{a}.1.net.0", hidden, dim)
        yield from _linear(f"{a}.1.net.2", dim, hidden)
        yield from _layernorm(f"{a}.2", dim)
        yield from _layernorm(f"{a}.3", dim)


def _swin_block(prefix: str, dim: int, time_dim: int, ratio: float, cfg: AuroraConfig):
    yield from _linear(f"{prefix}.norm1.ln_modulation.1", 2 * dim, time_dim, init="zeros")
    yield from _linear(f"{prefix}.attn.qkv", 3 * dim, dim)
    yield from _linear(f"{prefix}.attn.proj", dim, dim)
    if cfg.use_lora:
        for which, n_out in (("lora_proj", dim), ("lora_qkv", 3 * dim)):
            for k in range(cfg.num_lora_sets):
                lp = f"{prefix}.attn.{which}.loras.{k}"
                yield ParamSpec(f"{lp}.lora_A", (LORA_RANK, dim), "kaiming5", dim)
                yield ParamSpec(f"{lp}.lora_B", (n_out, LORA_RANK), "zeros")
    yield from _linear(f"{prefix}.norm2.ln_modulation.1", 2 * dim, time_dim, init="zeros")
    hidden = int(dim * ratio)
    yield from _linear(f"{prefix}.mlp.fc1", hidden, dim)
    yield from _linear(f"{prefix}.mlp.fc2", dim, hidden)


def param_specs(cfg: AuroraConfig) -> Iterator[ParamSpec]:
    """Every parameter of the model for configuration `cfg`, in registration order."""
    D, P, T = cfg.embed_dim, cfg.patch_size, cfg.max_history_size

    # ---- encoder (Perceiver3DEncoder) ----
    e = "encoder"
    yield ParamSpec(f"{e}.atmos_latents", (cfg.latent_levels - 1, D), "trunc02")
    yield ParamSpec(f"{e}.surf_level_encoding", (D,), "trunc02")
    yield from _linear(f"{e}.surf_mlp.net.0", int(D * cfg.mlp_ratio), D)
    yield from _linear(f"{e}.surf_mlp.net.2", D, int(D * cfg.mlp_ratio))
    yield from _layernorm(f"{e}.surf_norm", D)
    for name in ("pos_embed", "scale_embed", "lead_time_embed", "absolute_time_embed",
                 "atmos_levels_embed"):
        yield from _linear(f"{e}.{name}", D, D)
    yield from _patch_embed(f"{e}.surf_token_embeds", cfg.enc_surf_vars, D, T, P)
    if cfg.level_condition:
        for lvl in cfg.level_condition:
            yield from _patch_embed(
                f"{e}.atmos_token_embeds.layers.{level_to_str(lvl)}", cfg.enc_atmos_vars, D, T, P
            )
    else:
        yield from _patch_embed(f"{e}.atmos_token_embeds", cfg.enc_atmos_vars, D, T, P)
    yield from _resampler(
        f"{e}.level_agg", D, cfg.enc_depth, D // cfg.num_heads, cfg.num_heads, cfg.mlp_ratio,
        cfg.stabilise_level_agg,
    )

    # ---- backbone (Swin3DTransformerBackbone) ----
    b = "backbone"
    yield from _linear(f"{b}.time_mlp.0", D, D)
    yield from _linear(f"{b}.time_mlp.2", D, D)
    n_enc, n_dec = len(cfg.encoder_depths), len(cfg.decoder_depths)
    for i, depth in enumerate(cfg.encoder_depths):
        dim = D * 2**i
        for j in range(depth):
            yield from _swin_block(f"{b}.encoder_layers.{i}.blocks.{j}", dim, D, cfg.mlp_ratio, cfg)
        if i < n_enc - 1:
            yield from _linear(f"{b}.encoder_layers.{i}.downsample.reduction", 2 * dim, 4 * dim,
                               bias=False)
            yield from _layernorm(f"{b}.encoder_layers.{i}.downsample.norm", 4 * dim)
    for i, depth in enumerate(cfg.decoder_depths):
        dim = D * 2 ** (n_dec - 1 - i)
        for j in range(depth):
            yield from _swin_block(f"{b}.decoder_layers.{i}.blocks.{j}", dim, D, cfg.mlp_ratio, cfg)
        if i < n_dec - 1:
            yield from _linear(f"{b}.decoder_layers.{i}.upsample.lin1", 2 * dim, dim, bias=False)
            yield from _linear(f"{b}.decoder_layers.{i}.upsample.lin2", dim // 2, dim // 2,
                               bias=False)
            yield from _layernorm(f"{b}.decoder_layers.{i}.upsample.norm", dim // 2)

    # ---- decoder (Perceiver3DDecoder) ----
    d, D2 = "decoder", 2 * D
    head_dim = D2 // cfg.num_heads
    yield from _resampler(f"{d}.level_decoder", D2, cfg.dec_depth, head_dim, cfg.num_heads,
                          cfg.dec_mlp_ratio, False)
    if cfg.dec_separate_perceiver:
        yield from _resampler(f"{d}.level_decoder_alternate", D2, cfg.dec_depth, head_dim,
                              cfg.num_heads, cfg.dec_mlp_ratio, False)
    for v in cfg.dec_surf_vars:
        yield from _linear(f"{d}.surf_heads.{v}", P * P, D2)
    for v in cfg.dec_atmos_vars:
        if cfg.level_condition:
            for lvl in cfg.level_condition:
                yield from _linear(f"{d}.atmos_heads.{v}.layers.{level_to_str(lvl)}", P * P, D2)
        else:
            yield from _linear(f"{d}.atmos_heads.{v}", P * P, D2)
    yield from _linear(f"{d}.atmos_levels_embed", D2, D2)


# ---- initialisers (same distributions as the reference; RNG streams are not matched) ----
def _trunc02(t: torch.Tensor, spec: ParamSpec) -> None:
    nn.init.trunc_normal_(t, std=0.02)


def _kaiming5(t: torch.Tensor, spec: ParamSpec) -> None:
    # kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    bound = 1.0 / math.sqrt(spec.fan_in)
    nn.init.uniform_(t, -bound, bound)


_INITS = {
    "trunc02": _trunc02,
    "zeros": lambda t, s: nn.init.zeros_(t),
    "ones": lambda t, s: nn.init.ones_(t),
    "half": lambda t, s: nn.init.constant_(t, 0.5),
    "kaiming5": _kaiming5,
    "uniform_fan": _kaiming5,
}


class ParamTree(nn.Module):
    """Nested parameter containers built from `ParamSpec`s; no forward."""

    def __init__(self) -> None:
        super().__init__()

    def _child(self, name: str) -> "ParamTree":
        if name not in self._modules:
            self.add_module(name, ParamTree())
        return self._modules[name]  # type: ignore[return-value]

    def declare(self, spec: ParamSpec, prefix_to_strip: str = "") -> None:
        path = spec.name[len(prefix_to_strip):].split(".")
        node = self
        for part in path[:-1]:
            node = node._child(part)
        p = nn.Parameter(torch.empty(spec.shape))
        with torch.no_grad():
            _INITS[spec.init](p, spec)
        node.register_parameter(path[-1], p)

    def forward(self, *args, **kwargs):  # pragma: no cover - containers are not callable
        raise RuntimeError(
            "aurora_amd parameter containers hold weights only; the compute runs in the "
            "HIP engine (call the top-level Aurora model)."
        )
