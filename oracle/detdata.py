"""Deterministic, platform-independent test data.  TEST INFRASTRUCTURE ONLY.

Golden vectors must be reproducible on the GPU box, where neither /root/reference nor the
RNG stream of the torch build that generated them is guaranteed.  Weights and inputs are
therefore derived from a counter-based hash (splitmix64 over `crc32(name) + index`)
implemented with numpy integer arithmetic only: the same bits everywhere.

`det_state_dict` fills every parameter of a schema -- including the ones the reference
initialises to zero (LoRA B, AdaLN modulation; lora.py:51, film.py:33-36), otherwise those
paths would go untested (the reference's own tests randomise them for the same reason,
tests/test_rollout.py:23-35).
"""

from __future__ import annotations

import zlib
from datetime import datetime, timezone
from typing import Mapping, Sequence

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def det_uniform(name: str, shape: Sequence[int], seed: int = 0) -> np.ndarray:
    """float64 array of `shape`, i.i.d.-looking uniform on [-1, 1), a pure function of its args."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = np.uint64(zlib.crc32(name.encode())) * np.uint64(0x100000001B3) + np.uint64(seed)
        bits = _splitmix64(_splitmix64(np.full(1, base, dtype=np.uint64))[0]
                           + np.arange(n, dtype=np.uint64))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (2.0 * u - 1.0).reshape(tuple(shape))


def _stable_columns(name: str, d: int) -> np.ndarray | None:
    """Mask of input features of pos_embed / scale_embed that are numerically well-conditioned.

    The reference computes patch midpoints and root areas in fp32 and expands them at wavelengths
    down to 0.01 deg / 1e-4 km (fourier.py:112-119, posencoding.py:92-110): one fp32 ulp of the
    input (a libm / vectorisation difference between CPU families) moves the shortest-wavelength
    features by O(1) rad.  Golden vectors must be reproducible on a different host CPU, so the test
    weights ignore those chaotic columns (wavelength < 30 deg, < 100 km); they are covered bitwise by
    the same-machine tests (tests/test_encodings.py, engine-vs-oracle GPU tests).
    """
    if name == "encoder.pos_embed.weight":
        lam = np.logspace(np.log10(0.01), np.log10(720.0), d // 4)
        return np.tile(lam >= 30.0, 4)  # [sin_lat | cos_lat | sin_lon | cos_lon]
    if name == "encoder.scale_embed.weight":
        lam = np.logspace(np.log10(1.0814085263058073e-04), np.log10(511207893.39581096), d // 2)
        return np.tile(lam >= 100.0, 2)  # [sin | cos]
    return None


def det_param(name: str, shape: Sequence[int], seed: int = 0) -> np.ndarray:
    """A plausible parameter value for the schema entry `name`."""
    u = det_uniform(name, shape, seed)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        keep = _stable_columns(name, shape[1])
        if keep is not None:
            u = u * keep[None, :]
        return u * np.sqrt(3.0 / fan_in)
    if name.endswith(".weight"):  # 1-D weights are LayerNorm gains
        return 1.0 + 0.1 * u
    if name.endswith("surf_level_encoding"):
        return 0.5 * u
    return 0.1 * u  # biases


def det_state_dict(specs, dtype=torch.float64, seed: int = 0) -> dict[str, torch.Tensor]:
    """`specs`: iterable of objects with `.name` and `.shape` (aurora_amd ParamSpec) or pairs."""
    out = {}
    for s in specs:
        name, shape = (s.name, s.shape) if hasattr(s, "name") else s
        out[name] = torch.from_numpy(det_param(name, tuple(shape), seed)).to(dtype)
    return out


def det_inputs(surf_vars: Sequence[str], static_vars: Sequence[str], atmos_vars: Sequence[str],
               B: int, T: int, H: int, W: int, levels: Sequence[float],
               locations: Mapping[str, float], scales: Mapping[str, float], seed: int = 1,
               positive: Sequence[str] = ()):
    """Raw-unit inputs whose normalised values are det-uniform on [-1, 1) (|.| for `positive`).

    Returns (surf, static, atmos, lat, lon, times) with fp64 tensors.
    lat = linspace(90, -90, H), lon = linspace(0, 360, W + 1)[:-1] (README.md:87-97 recipe).
    """
    def lvl_key(lv):
        v = round(float(lv), 3)
        return (str(int(v)) if v % 1 == 0 else str(v)).replace(".", "_")

    surf, static, atmos = {}, {}, {}
    for v in surf_vars:
        u = det_uniform(f"in.surf.{v}", (B, T, H, W), seed)
        if v in positive:
            u = np.abs(u)
        surf[v] = torch.from_numpy(u * scales[v] + locations[v])
    for v in static_vars:
        u = det_uniform(f"in.static.{v}", (H, W), seed)
        static[v] = torch.from_numpy(u * scales[v] + locations[v])
    for v in atmos_vars:
        u = det_uniform(f"in.atmos.{v}", (B, T, len(levels), H, W), seed)
        if v in positive:
            u = np.abs(u)
        loc = np.array([locations[f"{v}_{lvl_key(lv)}"] for lv in levels])[:, None, None]
        sc = np.array([scales[f"{v}_{lvl_key(lv)}"] for lv in levels])[:, None, None]
        atmos[v] = torch.from_numpy(u * sc + loc)
    lat = torch.linspace(90, -90, H, dtype=torch.float64)
    lon = torch.linspace(0, 360, W + 1, dtype=torch.float64)[:-1]
    # timezone-aware, so that `.timestamp()` does not depend on the host's TZ setting
    times = tuple(datetime(2020, 6, 1 + b, 12, 0, tzinfo=timezone.utc) for b in range(B))
    return surf, static, atmos, lat, lon, times


# Raw input variables of the ocean-wave variant: wind as speed + direction (`dwi`), which the model's
# batch transform splits into 10u_wave / 10v_wave (aurora.py:863-871 upstream).
WAVE_RAW_SURF = ("2t", "10u", "10v", "msl", "swh", "mwd", "mwp", "pp1d", "shww", "mdww", "mpww", "shts", "mdts",
                 "mpts", "swh1", "mwd1", "mwp1", "swh2", "mwd2", "mwp2", "wind", "dwi")
_WAVE_HEIGHTS = ("swh", "shww", "shts", "swh1", "swh2")
_WAVE_ANGLES = ("mwd", "mdww", "mdts", "mwd1", "mwd2", "dwi")


def det_wave_inputs(static_vars: Sequence[str], atmos_vars: Sequence[str], B: int, T: int, H: int, W: int,
                    levels: Sequence[float], locations: Mapping[str, float], scales: Mapping[str, float],
                    seed: int = 1):
    """`det_inputs` for AuroraWave: strictly positive wave parameters, directions in [0, 360) degrees,
    and about a fifth of every wave system's heights exactly zero (absent system -> NaN marking)."""
    loc = dict(locations, dwi=0.0)
    sc = dict(scales, dwi=1.0)
    surf, static, atmos, lat, lon, times = det_inputs(WAVE_RAW_SURF, static_vars, atmos_vars, B, T, H, W, levels,
                                                      loc, sc, seed)
    for v in WAVE_RAW_SURF[4:]:
        u = det_uniform(f"in.surf.{v}", (B, T, H, W), seed)
        if v in _WAVE_ANGLES:
            x = (u + 1.0) * 180.0
        else:
            x = (u + 1.1) * sc[v]
            if v in _WAVE_HEIGHTS:
                x = np.where(det_uniform(f"in.absent.{v}", (B, T, H, W), seed) < -0.6, 0.0, x)
        surf[v] = torch.from_numpy(x)
    return surf, static, atmos, lat, lon, times
