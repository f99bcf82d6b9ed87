"""CPU oracle for the Aurora forward / rollout hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, op by op, the algorithm of the reference's
`Aurora.forward` (aurora/model/aurora.py:265-392) as plain functional torch-CPU code
over a flat `state_dict`.  It is the checker that the HIP engine in `aurora_amd/` is
compared against; nothing in the product path may import it (only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg do).

Pinning: tools/make_golden.py imports the real reference (through tools/ref_stub), runs its
`rollout()` in fp64 on deterministic weights / inputs (oracle/detdata.py) and stores the outputs as
tests/golden/*.npz; tests/test_oracle_golden.py checks this file against those vectors everywhere
(deviation <= 8e-15 where they were made).  tools/time_reference.py additionally compares the two
on the full 0.25-degree workload where /root/reference exists (7e-7 in fp32).

Everything is written for an arbitrary floating dtype (fp64 for tight checks, fp32
like-for-like).  `autocast=True` wraps the backbone in `torch.autocast("cpu", bf16)`
exactly where the reference does (aurora.py:327-343).

The functions take `cfg`, any object exposing the attributes of
`aurora_amd.model.schema.AuroraConfig`, a state dict `sd` and plain dict / tensor inputs.
"""

from __future__ import annotations

import contextlib
import math
from datetime import datetime, timedelta
from typing import Mapping, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------
def level_key(level) -> str:
    """normalisation.py:19-32."""
    v = round(float(level), 3)
    return (str(int(v)) if v % 1 == 0 else str(v)).replace(".", "_")


def linear(sd, prefix: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[f"{prefix}.weight"], sd.get(f"{prefix}.bias"))


def layer_norm(sd, prefix: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, x.shape[-1:], sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], eps)


# ----------------------------------------------------------------------------------------
# Fourier expansions (aurora/model/fourier.py:21-126, aurora/area.py:12-48)
# ----------------------------------------------------------------------------------------
RADIUS_EARTH_KM = 6378137 / 1000


def polygon_area(polygon: Tensor) -> Tensor:
    """Spherical polygon area in km^2 (area.py:12-48).  `polygon`: (n, 2) lat/lon degrees.

    Note the reference "closes" the loop by repeating the LAST vertex (area.py:30); that
    quirk is part of the published constant and is kept.
    """
    pts = torch.cat((polygon, polygon[-1:, :]), dim=0)
    n = pts.shape[0]
    total = torch.zeros((), dtype=pts.dtype)
    if n > 2:
        for i in range(n):
            lon_lo = torch.deg2rad(pts[i, 1])
            lat_mid = torch.deg2rad(pts[(i + 1) % n, 0])
            lon_hi = torch.deg2rad(pts[(i + 2) % n, 1])
            total = total + (lon_hi - lon_lo) * torch.sin(lat_mid)
    return torch.abs(total * RADIUS_EARTH_KM * RADIUS_EARTH_KM / 2)


_DELTA = 0.01
MIN_PATCH_AREA = polygon_area(
    torch.tensor(
        [[90, 0], [90, _DELTA], [90 - _DELTA, _DELTA], [90 - _DELTA, 0]], dtype=torch.float64
    )
).item()
AREA_EARTH = 4 * np.pi * RADIUS_EARTH_KM * RADIUS_EARTH_KM

# (lower wavelength, upper wavelength, assert_range) -- fourier.py:112-126
EXPANSIONS = {
    "pos": (_DELTA, 720.0, True),
    "scale": (MIN_PATCH_AREA, AREA_EARTH, True),
    "lead_time": (1 / 60, 24 * 7 * 3, True),
    "levels": (0.01, 1e5, True),
    "absolute_time": (1.0, 24 * 365.25, False),
}


def fourier_expansion(kind: str, x: Tensor, d: int) -> Tensor:
    """sin/cos features at d/2 log-spaced wavelengths, computed in fp64, returned as fp32."""
    lower, upper, check = EXPANSIONS[kind]
    # Range assertion, operator for operator as the reference writes it (fourier.py:64-72):
    # note `torch.all(x.abs() <= upper)` is a scalar there.
    ok = torch.logical_and(lower <= x.abs(), torch.all(x.abs() <= upper))
    if check and not bool(torch.all(torch.logical_or(ok, x == 0))):
        raise AssertionError(f"The input tensor is not within the configured range [{lower}, {upper}].")
    if d % 2:
        raise ValueError("The dimensionality must be a multiple of two.")
    x = x.double()
    wavelengths = torch.logspace(math.log10(lower), math.log10(upper), d // 2, base=10,
                                 dtype=torch.float64)
    prod = x[..., None] * (2 * np.pi / wavelengths)
    return torch.cat((prod.sin(), prod.cos()), dim=-1).float()


# ----------------------------------------------------------------------------------------
# position / scale encodings (aurora/model/posencoding.py:17-192)
# ----------------------------------------------------------------------------------------
def pos_scale_encodings(dim: int, lat: Tensor, lon: Tensor, patch: int) -> tuple[Tensor, Tensor]:
    """(L, dim) position and scale encodings of the patch grid; lat/lon fp32 vectors or matrices."""
    if lat.dim() == 1 and lon.dim() == 1:
        glat = lat[:, None].expand(-1, lon.shape[0])
        glon = lon[None, :].expand(lat.shape[0], -1)
    elif lat.dim() == 2 and lon.dim() == 2:
        glat, glon = lat, lon
    else:
        raise ValueError("Latitudes and longitudes must either both be vectors or both be matrices.")
    glat, glon = glat[None, None].contiguous(), glon[None, None].contiguous()
    k = (patch, patch)
    mid_lat = F.avg_pool2d(glat, k)[0, 0]
    mid_lon = F.avg_pool2d(glon, k)[0, 0]
    lat_max, lat_min = F.max_pool2d(glat, k)[0, 0], -F.max_pool2d(-glat, k)[0, 0]
    lon_max, lon_min = F.max_pool2d(glon, k)[0, 0], -F.max_pool2d(-glon, k)[0, 0]
    assert (lat_max > lat_min).all() and (lon_max > lon_min).all()
    # Rectangle-on-a-sphere area, R = 6371 km here (posencoding.py:48-53), then the root.
    area = (
        6371**2
        * torch.pi
        * (torch.sin(torch.deg2rad(lat_max)) - torch.sin(torch.deg2rad(lat_min)))
        * (torch.deg2rad(lon_max) - torch.deg2rad(lon_min))
    )
    assert (area > 0).all()
    root_area = torch.sqrt(area)
    assert dim % 4 == 0
    enc_lat = fourier_expansion("pos", mid_lat.reshape(-1), dim // 2)
    enc_lon = fourier_expansion("pos", mid_lon.reshape(-1), dim // 2)
    pos = torch.cat((enc_lat, enc_lon), dim=-1)
    scale = fourier_expansion("scale", root_area.reshape(-1), dim)
    return pos, scale


# ----------------------------------------------------------------------------------------
# patch embedding (aurora/model/patchembed.py:79-118)
# ----------------------------------------------------------------------------------------
def patch_embed(sd, prefix: str, x: Tensor, var_names: Sequence[str], patch: int) -> Tensor:
    """x: (B, V, T, H, W) -> (B, L, D).  Per-variable kernels concatenated, one strided conv."""
    B, V, T, H, W = x.shape
    assert len(var_names) == V and H % patch == 0 and W % patch == 0
    w = torch.cat([sd[f"{prefix}.weights.{n}"][:, :, :T] for n in var_names], dim=1)
    out = F.conv3d(x, w, sd[f"{prefix}.bias"], stride=(T, patch, patch))
    return out.reshape(B, w.shape[0], -1).transpose(1, 2)


# ----------------------------------------------------------------------------------------
# Perceiver blocks (aurora/model/perceiver.py:67-233)
# ----------------------------------------------------------------------------------------
def perceiver_attention(sd, prefix: str, latents: Tensor, ctx: Tensor, heads: int) -> Tensor:
    q = linear(sd, f"{prefix}.to_q", latents)
    k, v = linear(sd, f"{prefix}.to_kv", ctx).chunk(2, dim=-1)
    if f"{prefix}.ln_k.weight" in sd:  # layer norm before the head split (perceiver.py:144-147)
        k = layer_norm(sd, f"{prefix}.ln_k", k)
        q = layer_norm(sd, f"{prefix}.ln_q", q)

    def split(t):
        b, n, _ = t.shape
        return t.reshape(b, n, heads, -1).transpose(1, 2)

    out = F.scaled_dot_product_attention(split(q), split(k), split(v))
    out = out.transpose(1, 2).reshape(latents.shape[0], latents.shape[1], -1)
    return linear(sd, f"{prefix}.to_out", out)


def perceiver_resampler(sd, prefix: str, latents: Tensor, ctx: Tensor, depth: int, heads: int,
                        eps: float) -> Tensor:
    for i in range(depth):
        p = f"{prefix}.layers.{i}"
        latents = layer_norm(sd, f"{p}.2", perceiver_attention(sd, f"{p}.0", latents, ctx, heads),
                             eps) + latents
        hidden = F.gelu(linear(sd, f"{p}.1.net.0", latents))
        latents = layer_norm(sd, f"{p}.3", linear(sd, f"{p}.1.net.2", hidden), eps) + latents
    return latents


# ----------------------------------------------------------------------------------------
# encoder (aurora/model/encoder.py:173-366)
# ----------------------------------------------------------------------------------------
def dynamic_fields(times: Sequence[datetime], T: int, H: int, W: int, dtype) -> Tensor:
    """(B, T, 6, H, W) time-of-day / day-of-week / day-of-year planes (encoder.py:226-246)."""
    rows = []
    for t in times:
        vals = [
            np.cos(2 * np.pi * t.hour / 24), np.sin(2 * np.pi * t.hour / 24),
            np.cos(2 * np.pi * t.weekday() / 7), np.sin(2 * np.pi * t.weekday() / 7),
            np.cos(2 * np.pi * t.day / 365.25), np.sin(2 * np.pi * t.day / 365.25),
        ]
        ones = torch.ones((1, T, 1, H, W), dtype=dtype)
        rows.append(torch.cat([ones * v for v in vals], dim=-3))
    return torch.cat(rows, dim=0)


def encoder_forward(sd, cfg, surf: Mapping[str, Tensor], static: Mapping[str, Tensor],
                    atmos: Mapping[str, Tensor], lat: Tensor, lon: Tensor,
                    times: Sequence[datetime], levels: Sequence[float]) -> Tensor:
    """Inputs are normalised; static vars already carry (B, T, H, W).  Returns (B, L', D)."""
    D, P = cfg.embed_dim, cfg.patch_size
    surf_names, static_names, atmos_names = tuple(surf), tuple(static), tuple(atmos)
    x_surf = torch.stack(tuple(surf.values()), dim=2)      # (B, T, Vs, H, W)
    x_static = torch.stack(tuple(static.values()), dim=2)  # (B, T, Vst, H, W)
    x_atmos = torch.stack(tuple(atmos.values()), dim=2)    # (B, T, Va, C, H, W)
    B, T, _, C, H, W = x_atmos.shape
    dtype = x_surf.dtype

    if cfg.dynamic_vars:
        x_dyn = dynamic_fields(times, T, H, W, dtype)
        dyn_names = ("tod_cos", "tod_sin", "dow_cos", "dow_sin", "doy_cos", "doy_sin")
        x_surf = torch.cat((x_surf, x_static, x_dyn), dim=2)
        surf_names = surf_names + static_names + dyn_names
        if cfg.atmos_static_vars:
            atmos_names += tuple(f"static_{v}" for v in static_names + dyn_names)
            rep = lambda t: t[:, :, :, None].expand(-1, -1, -1, C, -1, -1)  # noqa: E731
            x_atmos = torch.cat((x_atmos, rep(x_static), rep(x_dyn)), dim=2)
    else:
        x_surf = torch.cat((x_surf, x_static), dim=2)
        surf_names = surf_names + static_names
        if cfg.atmos_static_vars:
            atmos_names = atmos_names + static_names
            x_atmos = torch.cat(
                (x_atmos, x_static[:, :, :, None].expand(-1, -1, -1, C, -1, -1)), dim=2)

    lat, lon = lat.float(), lon.float()
    assert lat.shape[0] == H and lon.shape[-1] == W

    xs = patch_embed(sd, "encoder.surf_token_embeds", x_surf.transpose(1, 2), surf_names, P)

    if cfg.simulate_indexing_bug and "z" in atmos_names:  # encoder.py:293-303
        iz, isz = atmos_names.index("z"), atmos_names.index("static_z")
        x_atmos = torch.cat((x_atmos[:, :, :isz], x_atmos[:, :, iz:iz + 1], x_atmos[:, :, isz + 1:]),
                            dim=2)

    # (B, T, V, C, H, W) -> per level (B, V, T, H, W)
    per_level = x_atmos.permute(0, 3, 2, 1, 4, 5)  # (B, C, V, T, H, W)
    if cfg.level_condition:
        xa = torch.stack(
            [patch_embed(sd, f"encoder.atmos_token_embeds.layers.{level_key(lv)}",
                         per_level[:, i], atmos_names, P) for i, lv in enumerate(levels)], dim=1)
    else:
        xa = patch_embed(sd, "encoder.atmos_token_embeds", per_level.reshape(B * C, *per_level.shape[2:]),
                         atmos_names, P).reshape(B, C, -1, D)

    xs = xs + sd["encoder.surf_level_encoding"][None, None, :].to(dtype)
    hidden = F.gelu(linear(sd, "encoder.surf_mlp.net.0", xs))
    xs = xs + layer_norm(sd, "encoder.surf_norm", linear(sd, "encoder.surf_mlp.net.2", hidden))

    lv_enc = fourier_expansion("levels", torch.tensor(levels), D).to(dtype)
    xa = xa + linear(sd, "encoder.atmos_levels_embed", lv_enc)[None, :, None, :]

    # Level aggregation: per grid column, (latent_levels-1) latent queries over C level keys.
    L = xa.shape[2]
    ctx = xa.permute(0, 2, 1, 3).reshape(B * L, C, D)
    lat_q = sd["encoder.atmos_latents"].to(dtype)[None].expand(B * L, -1, -1)
    agg = perceiver_resampler(sd, "encoder.level_agg", lat_q, ctx, cfg.enc_depth, cfg.num_heads,
                              cfg.perceiver_ln_eps)
    agg = agg.reshape(B, L, -1, D).permute(0, 2, 1, 3)  # (B, C', L, D)

    x = torch.cat((xs[:, None], agg), dim=1)
    pos, scale = pos_scale_encodings(D, lat, lon, P)
    x = x + linear(sd, "encoder.pos_embed", pos.to(dtype))[None, None]
    x = x + linear(sd, "encoder.scale_embed", scale.to(dtype))[None, None]
    x = x.reshape(B, -1, D)

    hours = cfg.timestep.total_seconds() / 3600
    lead = fourier_expansion("lead_time", hours * torch.ones(B, dtype=dtype), D).to(dtype)
    x = x + linear(sd, "encoder.lead_time_embed", lead)[:, None]
    stamps = torch.tensor([t.timestamp() / 3600 for t in times], dtype=torch.float32)
    absolute = fourier_expansion("absolute_time", stamps, D).to(dtype)
    x = x + linear(sd, "encoder.absolute_time_embed", absolute)[:, None]
    return x


# ----------------------------------------------------------------------------------------
# 3D Swin backbone (aurora/model/swin3d.py)
# ----------------------------------------------------------------------------------------
def adjust_windows(ws, ss, res):
    """util.py:53-71: clamp window to the grid and drop the shift on clamped axes."""
    ws, ss = list(ws), list(ss)
    for i in range(3):
        if res[i] <= ws[i]:
            ws[i], ss[i] = res[i], 0
    return tuple(ws), tuple(ss)


def two_sided(pad: int) -> tuple[int, int]:
    """Front/back split of a padding amount, front = pad // 2 (swin3d.py:177-194,250-269)."""
    return (pad // 2, pad - pad // 2) if pad else (0, 0)


def pad_chw(x: Tensor, pad: tuple[int, int, int], value: float = 0.0) -> Tensor:
    """Two-sided constant padding of the C, H, W axes of (B, C, H, W, D) (swin3d.py:272-275)."""
    (cf, cb), (ht, hb), (wl, wr) = two_sided(pad[0]), two_sided(pad[1]), two_sided(pad[2])
    return F.pad(x, (0, 0, wl, wr, ht, hb, cf, cb), value=value)


def crop_chw(x: Tensor, pad: tuple[int, int, int]) -> Tensor:
    (cf, cb), (ht, hb), (wl, wr) = two_sided(pad[0]), two_sided(pad[1]), two_sided(pad[2])
    _, C, H, W, _ = x.shape
    return x[:, cf:C - cb, ht:H - hb, wl:W - wr]


def to_windows(x: Tensor, ws) -> Tensor:
    """(B, C, H, W, D) -> (B*nW, Wc*Wh*Ww, D), windows ordered (b, c1, h1, w1) (swin3d.py:197-214)."""
    B, C, H, W, D = x.shape
    x = x.reshape(B, C // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], D)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], D)


def from_windows(w: Tensor, ws, B: int, C: int, H: int, W: int) -> Tensor:
    D = w.shape[-1]
    x = w.reshape(B, C // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], D)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, C, H, W, D)


def shift_mask(C: int, H: int, W: int, ws, ss, dtype) -> Tensor:
    """Additive 0 / -100 mask per window for shifted blocks (swin3d.py:288-360).

    The rolled grid is labelled with 27 communication groups (3 slices per axis); because
    longitude wraps, the two trailing W slices are merged; padding forms group 27.
    """
    label = torch.zeros((1, C, H, W, 1), dtype=dtype)
    slices = lambda w_, s_: (slice(0, -w_), slice(-w_, -s_), slice(-s_, None))  # noqa: E731
    cnt = 0
    for c in slices(ws[0], ss[0]):
        for h in slices(ws[1], ss[1]):
            for w in slices(ws[2], ss[2]):
                label[:, c, h, w, :] = cnt
                cnt += 1
    for base in (0, 9, 18):
        for a in (1, 4, 7):
            label = label.masked_fill(label == a + base, a + 1 + base)
    pad = ((-C) % ws[0], (-H) % ws[1], (-W) % ws[2])
    label = pad_chw(label, pad, value=cnt)
    g = to_windows(label, ws)[..., 0]  # (nW, N)
    diff = g[:, None, :] - g[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def lora_delta(sd, prefix: str, x: Tensor, step: int, cfg):
    """lora.py:56-63,105-129: rank-8 additive term, selected by roll-out step, off from lora_steps on."""
    if not cfg.use_lora or step >= cfg.lora_steps:
        return 0
    if cfg.lora_mode == "single":
        k = 0
    elif cfg.lora_mode == "from_second":
        if step == 0:
            return 0
        k = 0
    elif cfg.lora_mode == "all":
        k = step
    else:
        raise ValueError(f"Invalid mode: {cfg.lora_mode}")
    a, b = sd[f"{prefix}.loras.{k}.lora_A"], sd[f"{prefix}.loras.{k}.lora_B"]
    return (x @ a.transpose(0, 1) @ b.transpose(0, 1)) * (8 / 8)  # alpha / r, both 8


def window_attention(sd, prefix: str, xw: Tensor, mask, heads: int, step: int, cfg) -> Tensor:
    """swin3d.py:136-171.  xw: (nW*B, N, D); mask: (nW, N, N) or None."""
    n, N, D = xw.shape
    qkv = linear(sd, f"{prefix}.qkv", xw) + lora_delta(sd, f"{prefix}.lora_qkv", xw, step, cfg)
    qkv = qkv.reshape(n, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if mask is not None:
        reps = n // mask.shape[0]
        m = mask[None, :, None].repeat(reps, 1, 1, 1, 1).reshape(-1, 1, N, N)
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=m)
    else:
        out = F.scaled_dot_product_attention(q, k, v)
    out = out.transpose(1, 2).reshape(n, N, D)
    return linear(sd, f"{prefix}.proj", out) + lora_delta(sd, f"{prefix}.lora_proj", out, step, cfg)


def ada_layer_norm(sd, prefix: str, x: Tensor, c: Tensor) -> Tensor:
    """film.py:38-49 with scale_bias = 0: LN(x) * scale + shift."""
    shift, scale = linear(sd, f"{prefix}.ln_modulation.1", F.silu(c))[:, None].chunk(2, dim=-1)
    return F.layer_norm(x, x.shape[-1:]) * scale + shift


def swin_block(sd, prefix: str, x: Tensor, c: Tensor, res, heads: int, shifted: bool, step: int,
               cfg) -> Tensor:
    """swin3d.py:440-509."""
    C, H, W = res
    B, L, D = x.shape
    assert L == C * H * W
    base_ss = tuple(w // 2 for w in cfg.window_size) if shifted else (0, 0, 0)
    ws, ss = adjust_windows(cfg.window_size, base_ss, res)
    shortcut = x
    g = x.reshape(B, C, H, W, D)
    if any(ss):
        g = torch.roll(g, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        mask = shift_mask(C, H, W, ws, ss, x.dtype)
    else:
        mask = None
    pad = ((-C) % ws[0], (-H) % ws[1], (-W) % ws[2])
    g = pad_chw(g, pad)
    Cp, Hp, Wp = g.shape[1:4]
    a = window_attention(sd, f"{prefix}.attn", to_windows(g, ws), mask, heads, step, cfg)
    g = crop_chw(from_windows(a, ws, B, Cp, Hp, Wp), pad)
    if any(ss):
        g = torch.roll(g, shifts=ss, dims=(1, 2, 3))
    x = shortcut + ada_layer_norm(sd, f"{prefix}.norm1", g.reshape(B, L, D), c)
    hidden = F.gelu(linear(sd, f"{prefix}.mlp.fc1", x))
    return x + ada_layer_norm(sd, f"{prefix}.norm2", linear(sd, f"{prefix}.mlp.fc2", hidden), c)


def patch_merge(sd, prefix: str, x: Tensor, res) -> Tensor:
    """swin3d.py:526-555: 2x2 (H, W) gather in (h, w, D) order -> LN(4D) -> Linear(4D, 2D)."""
    C, H, W = res
    B, L, D = x.shape
    g = pad_chw(x.reshape(B, C, H, W, D), (0, H % 2, W % 2))
    H2, W2 = g.shape[2] // 2, g.shape[3] // 2
    g = g.reshape(B, C, H2, 2, W2, 2, D).permute(0, 1, 2, 4, 3, 5, 6).reshape(B, C * H2 * W2, 4 * D)
    return F.linear(layer_norm(sd, f"{prefix}.norm", g), sd[f"{prefix}.reduction.weight"])


def patch_split(sd, prefix: str, x: Tensor, res, crop) -> Tensor:
    """swin3d.py:574-613: Linear(D, 2D) -> 2x2 pixel shuffle -> crop -> LN(D/2) -> Linear."""
    C, H, W = res
    B, L, D = x.shape
    y = F.linear(x, sd[f"{prefix}.lin1.weight"])  # (B, L, 2D)
    D2 = y.shape[-1]
    g = y.reshape(B, C, H, W, 2, 2, D2 // 4).permute(0, 1, 2, 4, 3, 5, 6)
    g = crop_chw(g.reshape(B, C, 2 * H, 2 * W, D2 // 4), crop)
    g = g.reshape(B, -1, D2 // 4)
    return F.linear(layer_norm(sd, f"{prefix}.norm", g), sd[f"{prefix}.lin2.weight"])


def stage_resolutions(patch_res, n_stages: int):
    """swin3d.py:868-882: per-stage (C, H, W) and the merge padding applied after each stage."""
    all_res, pads = [tuple(patch_res)], []
    for _ in range(1, n_stages):
        C, H, W = all_res[-1]
        pads.append((0, H % 2, W % 2))
        all_res.append((C, (H + H % 2) // 2, (W + W % 2) // 2))
    pads.append((0, 0, 0))
    return all_res, pads


def backbone_forward(sd, cfg, x: Tensor, patch_res, step: int) -> Tensor:
    """swin3d.py:884-936."""
    assert x.shape[1] == patch_res[0] * patch_res[1] * patch_res[2]
    assert patch_res[0] % cfg.window_size[0] == 0
    n_enc, n_dec = len(cfg.encoder_depths), len(cfg.decoder_depths)
    all_res, pads = stage_resolutions(patch_res, n_enc)
    hours = cfg.timestep / timedelta(hours=1)
    lead = fourier_expansion("lead_time", hours * torch.ones(x.shape[0], dtype=torch.float32),
                             cfg.embed_dim).to(x.dtype)
    c = linear(sd, "backbone.time_mlp.2", F.silu(linear(sd, "backbone.time_mlp.0", lead)))

    skips = []
    for i, depth in enumerate(cfg.encoder_depths):
        p = f"backbone.encoder_layers.{i}"
        for j in range(depth):
            x = swin_block(sd, f"{p}.blocks.{j}", x, c, all_res[i], cfg.encoder_num_heads[i],
                           j % 2 == 1, step, cfg)
        skips.append(x)
        if i < n_enc - 1:
            x = patch_merge(sd, f"{p}.downsample", x, all_res[i])
    for i, depth in enumerate(cfg.decoder_depths):
        p = f"backbone.decoder_layers.{i}"
        idx = n_dec - 1 - i
        for j in range(depth):
            x = swin_block(sd, f"{p}.blocks.{j}", x, c, all_res[idx], cfg.decoder_num_heads[i],
                           j % 2 == 1, step, cfg)
        if i < n_dec - 1:
            x = patch_split(sd, f"{p}.upsample", x, all_res[idx], pads[idx - 1])
        if 0 < i < n_dec - 1:
            x = x + skips[idx - 1]
        elif i == n_dec - 1:
            x = torch.cat([x, skips[0]], dim=-1)
    return x


# ----------------------------------------------------------------------------------------
# decoder (aurora/model/decoder.py:140-276, aurora/model/util.py:18-41)
# ----------------------------------------------------------------------------------------
def unpatchify(x: Tensor, V: int, H: int, W: int, P: int) -> Tensor:
    """(B, L, C, V*P*P) with V fastest -> (B, V, C, H, W)."""
    B, _, C, _ = x.shape
    x = x.reshape(B, H // P, W // P, C, P, P, V).permute(0, 6, 3, 1, 4, 2, 5)
    return x.reshape(B, V, C, H, W)


def decoder_forward(sd, cfg, x: Tensor, surf_names, atmos_names, levels, patch_res, H: int, W: int):
    """Returns ({name: (B, H, W)}, {name: (B, C, H, W)}) in normalised space."""
    P, D2 = cfg.patch_size, 2 * cfg.embed_dim
    surf_names = tuple(surf_names) + tuple(f"{n}_mod" for n in surf_names if n in cfg.modulation_heads)
    atmos_names = tuple(atmos_names) + tuple(f"{n}_mod" for n in atmos_names
                                              if n in cfg.modulation_heads)
    B = x.shape[0]
    Cl, Hp, Wp = patch_res
    x = x.reshape(B, Cl, Hp * Wp, D2).transpose(1, 2)  # (B, L, C', D)

    xs = torch.stack([linear(sd, f"decoder.surf_heads.{n}", x[:, :, :1]) for n in surf_names], dim=-1)
    surf = unpatchify(xs.reshape(*xs.shape[:3], -1), len(surf_names), H, W, P)[:, :, 0]

    lv_enc = fourier_expansion("levels", torch.tensor(levels), D2).to(x.dtype)
    queries = linear(sd, "decoder.atmos_levels_embed", lv_enc)  # (C_A, D2)
    L = x.shape[1]
    q = queries[None].expand(B * L, -1, -1)
    ctx = x[:, :, 1:].reshape(B * L, Cl - 1, D2)
    main = perceiver_resampler(sd, "decoder.level_decoder", q, ctx, cfg.dec_depth, cfg.num_heads,
                               cfg.perceiver_ln_eps).reshape(B, L, len(levels), D2)
    sep = tuple(cfg.separate_perceiver)
    if cfg.modulation_heads:
        sep += tuple(f"{n}_mod" for n in sep)
    alt = main
    if sep:
        alt = perceiver_resampler(sd, "decoder.level_decoder_alternate", q, ctx, cfg.dec_depth,
                                  cfg.num_heads, cfg.perceiver_ln_eps).reshape(B, L, len(levels), D2)

    outs = []
    for n in atmos_names:
        src = alt if n in sep else main
        if cfg.level_condition:
            outs.append(torch.stack(
                [linear(sd, f"decoder.atmos_heads.{n}.layers.{level_key(lv)}", src[:, :, i])
                 for i, lv in enumerate(levels)], dim=-2))
        else:
            outs.append(linear(sd, f"decoder.atmos_heads.{n}", src))
    xa = torch.stack(outs, dim=-1)
    atm = unpatchify(xa.reshape(*xa.shape[:3], -1), len(atmos_names), H, W, P)
    return ({n: surf[:, i] for i, n in enumerate(surf_names)},
            {n: atm[:, i] for i, n in enumerate(atmos_names)})


# ----------------------------------------------------------------------------------------
# full forward and rollout (aurora/model/aurora.py:265-392, aurora/rollout.py:14-49)
# ----------------------------------------------------------------------------------------
_DIFF_HISTORY = {"pm1": 0, "pm2p5": 0, "pm10": 0, "co": 1, "tcco": 1, "no": 0, "tc_no": 0,
                 "no2": 0, "tcno2": 0, "so2": 1, "tcso2": 1, "go3": 1, "gtco3": 1}


def _norm_surf(x, name, stats, locations, scales, inverse=False):
    loc, sc = stats[name] if (stats and name in stats) else (locations[name], scales[name])
    return x * sc + loc if inverse else (x - loc) / sc


def _norm_atmos(x, name, levels, locations, scales, inverse=False):
    loc = torch.tensor([locations[f"{name}_{level_key(lv)}"] for lv in levels], dtype=x.dtype)
    sc = torch.tensor([scales[f"{name}_{level_key(lv)}"] for lv in levels], dtype=x.dtype)
    loc, sc = loc[:, None, None], sc[:, None, None]
    return x * sc + loc if inverse else (x - loc) / sc


def _pollution_pre(sd, kind: str, name: str, z: Tensor) -> Tensor:
    """aurora.py:726-759: clamp / log feature pair through a Linear(2, 1)."""
    eps = 1e-4
    feats = torch.stack([z.clamp(min=0, max=2.5),
                         (torch.log(z.clamp(min=eps)) - np.log(eps)) / (-np.log(eps))], dim=-1)
    return linear(sd, f"{kind}_feature_combiner.{name}", feats)[..., 0]


WAVE_DENSITY_VARS = ("swh", "mwd", "mwp", "pp1d", "shww", "mdww", "mpww", "shts", "mdts", "mpts", "swh1", "mwd1",
                     "mwp1", "swh2", "mwd2", "mwp2", "wind", "10u_wave", "10v_wave")   # aurora.py:819-822
WAVE_ANGLE_VARS = ("mwd", "mdww", "mdts", "mwd1", "mwd2")                              # aurora.py:823


def wave_batch_transform(surf: Mapping[str, Tensor], rollout_step: int) -> dict:
    """aurora.py:854-890: wind speed/direction -> components; absent wave systems -> NaN (raw data only)."""
    surf = dict(surf)
    if "dwi" in surf and "wind" in surf:
        surf["10u_wave"] = -surf["wind"] * torch.sin(torch.deg2rad(surf["dwi"]))
        surf["10v_wave"] = -surf["wind"] * torch.cos(torch.deg2rad(surf["dwi"]))
        del surf["dwi"]
    if rollout_step == 0:
        for height, others in (("swh", ("mwd", "mwp", "pp1d")), ("shww", ("mdww", "mpww")),
                               ("shts", ("mdts", "mdts")), ("swh1", ("mwd1", "mwp1")), ("swh2", ("mwd2", "mwp2"))):
            gone = surf[height] < 1e-4
            if gone.sum() > 0:
                for name in (height,) + others:
                    x = surf[name].clone()
                    x[gone] = float("nan")
                    surf[name] = x
    return surf


def _wave_pre(surf: dict) -> dict:
    """aurora.py:892-912 on the normalised variables (the dictionary is extended while iterating a copy
    of its keys, so new channels go to the end in creation order)."""
    surf = dict(surf)
    for name in list(surf):
        x = surf[name]
        if name in WAVE_DENSITY_VARS and f"{name}_density" not in surf:
            surf[f"{name}_density"] = (~torch.isnan(x)).to(x.dtype)
            surf[name] = x.nan_to_num(0)
        if name in WAVE_ANGLE_VARS and not (f"{name}_sin" in surf and f"{name}_cos" in surf):
            surf[f"{name}_sin"] = torch.sin(torch.deg2rad(x)).nan_to_num(0)
            surf[f"{name}_cos"] = torch.cos(torch.deg2rad(x)).nan_to_num(0)
            del surf[name]
    return surf


def _wave_post(pred: dict, wmb: Tensor) -> dict:
    """aurora.py:914-941: angles back from sin/cos; values NaN where the density head says 'absent'
    or outside water bodies."""
    pred = dict(pred)
    water = wmb > 0
    for name in WAVE_ANGLE_VARS:
        if f"{name}_sin" in pred and f"{name}_cos" in pred:
            pred[name] = torch.rad2deg(torch.atan2(pred.pop(f"{name}_sin"), pred.pop(f"{name}_cos"))) % 360
    for name in WAVE_DENSITY_VARS:
        if name in pred:
            density = torch.sigmoid(pred.pop(f"{name}_density")) * water
            data = pred[name] * water
            data[density < 0.5] = float("nan")
            pred[name] = data
    return pred


def forward(sd: Mapping[str, Tensor], cfg, surf: Mapping[str, Tensor], static: Mapping[str, Tensor],
            atmos: Mapping[str, Tensor], lat: Tensor, lon: Tensor, times: Sequence[datetime],
            levels: Sequence[float], rollout_step: int, locations: Mapping[str, float],
            scales: Mapping[str, float], surf_stats=None, autocast: bool = False,
            variant: str = "base"):
    """One model step on raw (unnormalised) inputs.

    surf: name -> (B, T, H, W); static: name -> (H, W); atmos: name -> (B, T, C, H, W).
    Returns (surf_pred {name: (B, 1, H', W)}, atmos_pred {name: (B, 1, C, H', W)}, lat')
    in physical units, where H' drops the last latitude row if H % patch == 1
    (batch.py:142-168).  `variant` is "base", "air_pollution" (hooks aurora.py:726-796) or "wave"
    (hooks aurora.py:854-941).
    """
    dtype = next(iter(sd.values())).dtype
    P = cfg.patch_size
    cast = lambda d: {k: v.to(dtype) for k, v in d.items()}  # noqa: E731
    surf, static, atmos = cast(surf), cast(static), cast(atmos)
    lat, lon = lat.to(dtype), lon.to(dtype)
    if variant == "wave":
        surf = wave_batch_transform(surf, rollout_step)

    surf = {k: _norm_surf(v, k, surf_stats, locations, scales) for k, v in surf.items()}
    static = {k: _norm_surf(v, k, surf_stats, locations, scales) for k, v in static.items()}
    atmos = {k: _norm_atmos(v, k, levels, locations, scales) for k, v in atmos.items()}

    H, W = next(iter(surf.values())).shape[-2:]
    if W % P:
        raise ValueError("Width of the data must be a multiple of the patch size.")
    if H % P == 1:
        cut = lambda d: {k: v[..., :-1, :] for k, v in d.items()}  # noqa: E731
        surf, static, atmos, lat, H = cut(surf), cut(static), cut(atmos), lat[:-1], H - 1
    elif H % P:
        raise ValueError("There can at most be one latitude too many.")
    patch_res = (cfg.latent_levels, H // P, W // P)

    B, T = next(iter(surf.values())).shape[:2]
    static_bt = {k: v[None, None].repeat(B, T, 1, 1) for k, v in static.items()}

    enc_surf, enc_atmos = dict(surf), dict(atmos)
    for k in cfg.positive_surf_vars:
        if k in enc_surf:
            enc_surf[k] = enc_surf[k].clamp(min=0)
    for k in cfg.positive_atmos_vars:
        if k in enc_atmos:
            enc_atmos[k] = enc_atmos[k].clamp(min=0)
    if variant == "air_pollution":
        enc_surf = {k: _pollution_pre(sd, "surf", k, v) if k in cfg.positive_surf_vars else v
                    for k, v in enc_surf.items()}
        enc_atmos = {k: _pollution_pre(sd, "atmos", k, v) if k in cfg.positive_atmos_vars else v
                     for k, v in enc_atmos.items()}

    if variant == "wave":
        # (the reference mutates the batch the decoder sees, so the heads follow the expanded names)
        surf = enc_surf = _wave_pre(enc_surf)

    x = encoder_forward(sd, cfg, enc_surf, static_bt, enc_atmos, lat, lon, times, levels)
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()
    with ctx:
        x = backbone_forward(sd, cfg, x, patch_res, rollout_step)
    surf_pred, atmos_pred = decoder_forward(sd, cfg, x, tuple(surf), tuple(atmos), levels,
                                            patch_res, H, W)
    surf_pred = {k: v[:, None] for k, v in surf_pred.items()}
    atmos_pred = {k: v[:, None] for k, v in atmos_pred.items()}

    if variant == "air_pollution":  # aurora.py:761-796
        def diff(prev, model, name):
            if name in _DIFF_HISTORY:
                return model[name] + (1 + model[f"{name}_mod"]) * prev[name][:, _DIFF_HISTORY[name]]
            return model[name]

        # NB: broadcasting of (B, 1, ...) with (B, ...) follows the reference literally.
        surf_pred = {k: diff(surf, surf_pred, k) for k in surf}
        atmos_pred = {k: diff(atmos, atmos_pred, k) for k in atmos}
        if cfg.use_lora:
            parts = [atmos_pred["so2"][..., i, :, :].clamp(max=1) if lv >= 850
                     else atmos_pred["so2"][..., i, :, :] for i, lv in enumerate(levels)]
            atmos_pred["so2"] = torch.stack(parts, dim=-3)

    if variant == "wave":
        surf_pred = _wave_post(surf_pred, static["wmb"])

    new_step = rollout_step + 1
    if new_step >= 1 if cfg.clamp_at_first_step else new_step > 1:
        surf_pred = {k: v.clamp(min=0) if k in cfg.positive_surf_vars else v
                     for k, v in surf_pred.items()}
        atmos_pred = {k: v.clamp(min=0) if k in cfg.positive_atmos_vars else v
                      for k, v in atmos_pred.items()}

    surf_pred = {k: _norm_surf(v, k, surf_stats, locations, scales, True) for k, v in surf_pred.items()}
    atmos_pred = {k: _norm_atmos(v, k, levels, locations, scales, True) for k, v in atmos_pred.items()}
    return surf_pred, atmos_pred, lat


def rollout(sd, cfg, surf, static, atmos, lat, lon, times, levels, steps: int, locations, scales,
            **kw):
    """rollout.py:14-49 as a generator of (surf_pred, atmos_pred, times) per step."""
    dtype = next(iter(sd.values())).dtype
    P = cfg.patch_size
    surf = {k: v.to(dtype) for k, v in surf.items()}
    atmos = {k: v.to(dtype) for k, v in atmos.items()}
    static = {k: v.to(dtype) for k, v in static.items()}
    if kw.get("variant") == "wave":
        surf = wave_batch_transform(surf, 0)
    H = next(iter(surf.values())).shape[-2]
    if H % P == 1:
        cut = lambda d: {k: v[..., :-1, :] for k, v in d.items()}  # noqa: E731
        surf, static, atmos, lat = cut(surf), cut(static), cut(atmos), lat[:-1]
    times = tuple(times)
    for step in range(steps):
        sp, ap, _ = forward(sd, cfg, surf, static, atmos, lat, lon, times, levels, step,
                            locations, scales, **kw)
        times = tuple(t + cfg.timestep for t in times)
        yield sp, ap, times
        surf = {k: torch.cat([surf[k][:, 1:], v], dim=1) for k, v in sp.items()}
        atmos = {k: torch.cat([atmos[k][:, 1:], v], dim=1) for k, v in ap.items()}
