"""Per-variable normalisation statistics and the affine (un)normalisation helpers.

Host-side mirror of the reference's `aurora/normalisation.py` (level_to_str :19-32,
normalise_surf_var :34-48, normalise_atmos_var :51-70, tables :77-457).  The
statistics live in `data/norm_stats.json` (exported by tools/export_norm_stats.py);
`locations` and `scales` are plain module-level dicts that users may extend, exactly
like the reference (docs/finetuning.md:122-147).  The HIP engine re-reads them at
every step through :func:`surf_affine` / :func:`atmos_affine`, so user edits are seen.

These torch helpers serve the public `Batch.normalise()` / `Batch.unnormalise()`
utilities.  `Aurora.forward` does not call them: the engine fuses the affine maps
into its patch-embed load and its unpatchify store.
"""

from __future__ import annotations

import json
from pathlib import Path
from typing import Mapping, Optional, Sequence

import torch

__all__ = [
    "level_to_str",
    "normalise_surf_var",
    "normalise_atmos_var",
    "unnormalise_surf_var",
    "unnormalise_atmos_var",
    "surf_affine",
    "atmos_affine",
    "locations",
    "scales",
]

_stats = json.loads((Path(__file__).parent / "data" / "norm_stats.json").read_text())
locations: dict[str, float] = dict(_stats["locations"])
scales: dict[str, float] = dict(_stats["scales"])
del _stats


def level_to_str(level: float) -> str:
    """Canonical text form of a pressure level: `850`, `0_5` (reference :19-32)."""
    value = round(float(level), 3)
    text = str(int(value)) if value == int(value) else str(value)
    return text.replace(".", "_")


def surf_affine(
    name: str, stats: Optional[Mapping[str, tuple[float, float]]] = None
) -> tuple[float, float]:
    """(location, scale) of a surface-level or static variable; `stats` overrides the table."""
    if stats and name in stats:
        loc, scale = stats[name]
        return float(loc), float(scale)
    return locations[name], scales[name]  # KeyError for unknown names, like the reference.


def atmos_affine(name: str, atmos_levels: Sequence[float]) -> tuple[list[float], list[float]]:
    """Per-level (locations, scales) of an atmospheric variable."""
    keys = [f"{name}_{level_to_str(lvl)}" for lvl in atmos_levels]
    return [locations[k] for k in keys], [scales[k] for k in keys]


def normalise_surf_var(
    x: torch.Tensor,
    name: str,
    stats: Optional[Mapping[str, tuple[float, float]]] = None,
    unnormalise: bool = False,
) -> torch.Tensor:
    loc, scale = surf_affine(name, stats)
    return x * scale + loc if unnormalise else (x - loc) / scale


def normalise_atmos_var(
    x: torch.Tensor,
    name: str,
    atmos_levels: Sequence[float],
    unnormalise: bool = False,
) -> torch.Tensor:
    locs, scs = atmos_affine(name, atmos_levels)
    # Levels sit on dim -3 of (..., c, h, w).
    loc = torch.tensor(locs, dtype=x.dtype, device=x.device)[:, None, None]
    scale = torch.tensor(scs, dtype=x.dtype, device=x.device)[:, None, None]
    return x * scale + loc if unnormalise else (x - loc) / scale


def unnormalise_surf_var(x, name, stats=None):
    return normalise_surf_var(x, name, stats, unnormalise=True)


def unnormalise_atmos_var(x, name, atmos_levels):
    return normalise_atmos_var(x, name, atmos_levels, unnormalise=True)
