"""Raw-batch preparation of the ocean-wave variant (reference aurora/model/aurora.py:854-890).

Runs once on the user's raw batch (and idempotently at every `forward`), before normalisation:
splits wind speed/direction into components and marks absent wave systems with NaN.  This is input
preparation on the host API level (plain tensor indexing), not part of the accelerated step; the
in-step halves of the wave handling (density channels, sin/cos of angles, their inverse) are
transform codes of the patchify / unpatchify kernels (aurora_amd/engine/engine.py).
"""

from __future__ import annotations

import dataclasses

import torch

from aurora_amd.batch import Batch

# (significant-height variable, components that are undefined when the height is ~0).
# NB the reference lists ("mdts", "mdts") for the total swell, so `mpts` is never masked; kept.
_SYSTEMS = (
    ("swh", ("mwd", "mwp", "pp1d")),
    ("shww", ("mdww", "mpww")),
    ("shts", ("mdts", "mdts")),
    ("swh1", ("mwd1", "mwp1")),
    ("swh2", ("mwd2", "mwp2")),
)
_DIRECTIONS = {"mwd", "mdww", "mdts", "mwd1", "mwd2"}


def transform_batch(batch: Batch) -> Batch:
    surf = dict(batch.surf_vars)
    if "dwi" in surf and "wind" in surf:
        ang = torch.deg2rad(surf["dwi"])
        surf["10u_wave"] = -surf["wind"] * torch.sin(ang)
        surf["10v_wave"] = -surf["wind"] * torch.cos(ang)
        del surf["dwi"]
    if batch.metadata.rollout_step == 0:
        for height, others in _SYSTEMS:
            absent = surf[height] < 1e-4
            if bool(absent.any()):
                for name in (height,) + others:
                    x = surf[name].clone()
                    x[absent] = float("nan")
                    surf[name] = x
                    if name not in _DIRECTIONS:
                        assert int((x < 1e-4).sum()) == 0
    return dataclasses.replace(batch, surf_vars=surf)
