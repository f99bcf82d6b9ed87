"""Synthetic checkpoints in the PUBLISHED key layouts, shared by tools/make_compat_fixtures.py (which runs the
reference's adapters on them and stores the results) and tests/test_compat.py (which runs aurora_amd's adapters on the
same inputs).  Layouts read off the reference's adapters (aurora/model/compat.py:18-284): the adapters are the only
description of the published files that exists offline.

`old_layout(family, shapes, patch)` starts from the CURRENT schema (`shapes`: state_dict key -> shape of a tiny model
of that family) and re-creates the published layout: fused patch embeddings and heads, the `net.` prefix, the doubly
specified level-conditioned embeddings of the air-pollution file with their throw-away tensors, the `k_ln` / `q_ln`
names of the wave file.  Values are a pure function of the OLD key (oracle/detdata.py), so nothing but names is stored.
"""
from __future__ import annotations

import torch

from oracle import detdata
from tests.golden_cases import _TINY

ERA5_SURF = ("2t", "10u", "10v", "msl")
ERA5_STATIC = ("lsm", "z", "slt")
ERA5_ATMOS = ("z", "u", "v", "t", "q")
CAMS_SURF = ("pm1", "pm2p5", "pm10", "tcco", "tc_no", "tcno2", "gtco3", "tcso2")
CAMS_STATIC = ("static_ammonia", "static_ammonia_log", "static_co", "static_co_log",
               "static_nox", "static_nox_log", "static_so2", "static_so2_log")
CAMS_DYNAMIC = ("tod_cos", "tod_sin", "dow_cos", "dow_sin", "doy_cos", "doy_sin")
CAMS_ATMOS = ("co", "no", "no2", "go3", "so2")
CAMS_LEVELS = (50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000)

FAMILIES = {
    "pretrained": dict(cls="Aurora", kwargs=dict(**_TINY, use_lora=True), patch=4),
    "air_pollution": dict(cls="AuroraAirPollution", kwargs=dict(**_TINY), patch=3),
    "wave": dict(cls="AuroraWave", kwargs=dict(**_TINY), patch=4),
}


def digest(d: dict[str, torch.Tensor]) -> dict:
    """key -> [shape, CRC-32 of the contiguous float32 bytes]: exact equality of a state dict in a few kilobytes."""
    import zlib

    return {k: [list(v.shape), zlib.crc32(v.detach().float().contiguous().numpy().tobytes())] for k, v in d.items()}


def _t(name: str, shape) -> torch.Tensor:
    return torch.from_numpy(detdata.det_uniform(name, tuple(shape))).float()


def _fuse_embed(shapes: dict, out: dict, prefix: str, names, fused_key: str) -> None:
    """`{prefix}.weights.<name>` (D, 1, T, P, P) for `names` -> one (D, V, T, P, P) tensor under `fused_key`."""
    D, _, T, P, _ = shapes[f"{prefix}.weights.{names[0]}"]
    for n in names:
        shapes.pop(f"{prefix}.weights.{n}")
    out[fused_key] = _t(fused_key, (D, len(names), T, P, P))


def _fuse_head(shapes: dict, out: dict, per_var: str, names, fused: str, patch: int, E: int = 0) -> None:
    """`per_var.format(name).{weight,bias}` -> one Linear(E, P*P*V) under `fused` (V fastest in the output index).
    With `E` given nothing is removed from `shapes` (heads the adapter drops or renames)."""
    if not E:
        E = shapes[per_var.format(names[0]) + ".weight"][1]
        for n in names:
            shapes.pop(per_var.format(n) + ".weight")
            shapes.pop(per_var.format(n) + ".bias")
    out[f"{fused}.weight"] = _t(f"{fused}.weight", (len(names) * patch ** 2, E))
    out[f"{fused}.bias"] = _t(f"{fused}.bias", (len(names) * patch ** 2,))


def old_layout(family: str, shapes_new: dict, patch: int) -> dict[str, torch.Tensor]:
    shapes = dict(shapes_new)
    out: dict[str, torch.Tensor] = {}
    enc_s, enc_a = "encoder.surf_token_embeds", "encoder.atmos_token_embeds"

    if family == "pretrained":
        _fuse_embed(shapes, out, enc_s, ERA5_SURF + ERA5_STATIC, f"{enc_s}.weight")
        _fuse_embed(shapes, out, enc_a, ERA5_ATMOS, f"{enc_a}.weight")
        _fuse_head(shapes, out, "decoder.surf_heads.{}", ERA5_SURF, "decoder.surf_head", patch)
        _fuse_head(shapes, out, "decoder.atmos_heads.{}", ERA5_ATMOS, "decoder.atmos_head", patch)
        for k, shp in shapes.items():
            out[k] = _t(k, shp)
        return {f"net.{k}": v for k, v in out.items()}      # published ERA5 files carry a `net.` prefix

    if family == "wave":
        for k, shp in shapes.items():
            old = k.replace(".ln_k.", ".k_ln.").replace(".ln_q.", ".q_ln.")
            out[old] = _t(old, shp)
        return out

    assert family == "air_pollution"
    # -- encoder, surface level: ERA5 variables fused the ERA5 way, everything new under `weight_new`
    _fuse_embed(shapes, out, enc_s, ERA5_SURF + ERA5_STATIC, f"{enc_s}.weight")
    _fuse_embed(shapes, out, enc_s, CAMS_SURF + CAMS_STATIC + CAMS_DYNAMIC, f"{enc_s}.weight_new")
    # -- encoder, atmospheric levels.  Published: ONE shared ERA5 embedding (fused) + bias, which the adapter clones to
    # every level; a second, level-conditioned instance `_new` whose per-level `weight_new` holds the pollution
    # variables and whose bias is ADDED; the static / dynamic planes of every level in one `weight_new2`; plus
    # tensors that are simply dropped (`weight_new`, per-level `weight`, per-level `weight_new2`).
    D, _, T, P, _ = shapes[f"{enc_a}.layers.50.weights.u"]
    for lvl in CAMS_LEVELS:
        for n in (ERA5_ATMOS + CAMS_ATMOS + tuple(f"static_{m}" for m in ERA5_STATIC + CAMS_STATIC + CAMS_DYNAMIC)):
            shapes.pop(f"{enc_a}.layers.{lvl}.weights.{n}")
        shapes.pop(f"{enc_a}.layers.{lvl}.bias")
        out[f"{enc_a}_new.layers.{lvl}.weight"] = _t(f"{enc_a}_new.layers.{lvl}.weight", (D, 5, T, P, P))
        out[f"{enc_a}_new.layers.{lvl}.weight_new"] = _t(f"{enc_a}_new.layers.{lvl}.weight_new", (D, 5, T, P, P))
        out[f"{enc_a}_new.layers.{lvl}.weight_new2"] = _t(f"{enc_a}_new.layers.{lvl}.weight_new2", (D, 17, T, P, P))
        out[f"{enc_a}_new.layers.{lvl}.bias"] = _t(f"{enc_a}_new.layers.{lvl}.bias", (D,))
    out[f"{enc_a}.weight"] = _t(f"{enc_a}.weight", (D, 5, T, P, P))
    out[f"{enc_a}.bias"] = _t(f"{enc_a}.bias", (D,))
    out[f"{enc_a}.weight_new"] = _t(f"{enc_a}.weight_new", (D, 5, T, P, P))
    out[f"{enc_a}.weight_new2"] = _t(f"{enc_a}.weight_new2", (D, 17, T, P, P))
    # -- feature combiners: published for every variable, kept for the pollution variables only
    for kind, names in (("surf", ERA5_SURF), ("atmos", ERA5_ATMOS)):
        for n in names:
            out[f"{kind}_feature_combiner.{n}.weight"] = _t(f"{kind}_feature_combiner.{n}.weight", (1, 2))
            out[f"{kind}_feature_combiner.{n}.bias"] = _t(f"{kind}_feature_combiner.{n}.bias", (1,))
    # -- decoder: second Perceiver under its old name; fused heads
    for k in [k for k in shapes if k.startswith("decoder.level_decoder_alternate")]:
        shp = shapes.pop(k)
        old = "decoder.level_decoder_new" + k[len("decoder.level_decoder_alternate"):]
        out[old] = _t(old, shp)
    _fuse_head(shapes, out, "decoder.surf_heads.{}", ERA5_SURF, "decoder.surf_head", patch)
    _fuse_head(shapes, out, "decoder.surf_heads.{}", CAMS_SURF, "decoder.surf_head_new", patch)
    for n in CAMS_SURF:
        shapes.pop(f"decoder.surf_heads.{n}_mod.weight")
        shapes.pop(f"decoder.surf_heads.{n}_mod.bias")
    E = out["decoder.surf_head.weight"].shape[1]
    _fuse_head(shapes, out, "decoder.surf_heads.{}", ERA5_SURF + CAMS_SURF, "decoder.surf_head_mod", patch, E=E)
    for lvl in CAMS_LEVELS:
        for names, fused in ((ERA5_ATMOS, f"decoder.atmos_head.layers.{lvl}"),
                             (CAMS_ATMOS, f"decoder.atmos_head_new.layers.{lvl}")):
            for n in names:
                shapes.pop(f"decoder.atmos_heads.{n}.layers.{lvl}.weight")
                shapes.pop(f"decoder.atmos_heads.{n}.layers.{lvl}.bias")
            out[f"{fused}.weight"] = _t(f"{fused}.weight", (5 * patch ** 2, E))
            out[f"{fused}.bias"] = _t(f"{fused}.bias", (5 * patch ** 2,))
        for n in CAMS_ATMOS:
            shapes.pop(f"decoder.atmos_heads.{n}_mod.layers.{lvl}.weight")
            shapes.pop(f"decoder.atmos_heads.{n}_mod.layers.{lvl}.bias")
        for fused in (f"decoder.atmos_head_mod.layers.{lvl}", f"decoder.atmos_head_mod_new.layers.{lvl}"):
            out[f"{fused}.weight"] = _t(f"{fused}.weight", (5 * patch ** 2, E))   # the first is dropped by the adapter
            out[f"{fused}.bias"] = _t(f"{fused}.bias", (5 * patch ** 2,))
    for k, shp in shapes.items():
        out[k] = _t(k, shp)
    return out
