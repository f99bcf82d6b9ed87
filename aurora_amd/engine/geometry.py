"""Host-side geometry of the 3D Swin backbone: stage resolutions and window token tables.

The reference expresses shifted-window attention as a chain of full-tensor copies
(roll -> pad -> partition -> attention -> reverse -> crop -> un-roll, swin3d.py:471-505) plus
a (nW, N, N) additive mask tensor (swin3d.py:303-360).  All of that is index arithmetic that
depends only on (C, H, W), the window size and the shift.  Here it is evaluated once on the
host, in closed form, into two small tables per (stage, shifted?):

  tok[w, n]  int32  token index (c*H + h)*W + w_ of the n-th position of window w in the
                    UN-rolled, UN-padded token order, or -1 for a zero-padded position
  grp[w, n]  uint8  communication-group label of that position (shifted blocks only)

The HIP attention kernel gathers and scatters through `tok` and masks through `grp`
(score -= 100 where labels differ), so no shuffle copy or mask tensor ever exists on the
device.  tests/test_geometry.py checks these tables against the roll/pad/partition chain.
"""

from __future__ import annotations

from functools import lru_cache

import numpy as np

Res = tuple[int, int, int]


def adjust_windows(ws: Res, ss: Res, res: Res) -> tuple[Res, Res]:
    """Clamp the window to the grid; a clamped axis is not shifted (reference util.py:53-71)."""
    ws_, ss_ = list(ws), list(ss)
    for i in range(3):
        if res[i] <= ws[i]:
            ws_[i], ss_[i] = res[i], 0
    return tuple(ws_), tuple(ss_)  # type: ignore[return-value]


def stage_resolutions(patch_res: Res, n_stages: int) -> tuple[list[Res], list[Res]]:
    """Per-stage (C, H, W) and the (0, pad_h, pad_w) merge padding after each stage
    (reference swin3d.py:868-882); the level axis is never merged."""
    res, pads = [tuple(patch_res)], []
    for _ in range(1, n_stages):
        C, H, W = res[-1]
        pads.append((0, H % 2, W % 2))
        res.append((C, (H + H % 2) // 2, (W + W % 2) // 2))
    pads.append((0, 0, 0))
    return res, pads  # type: ignore[return-value]


def _axis_labels(n: int, ws: int, ss: int) -> np.ndarray:
    """Slice label (0, 1, 2) of every coordinate of one rolled axis.

    The reference assigns slices [0, n-ws), [n-ws, n-ss), [n-ss, n) in that order
    (swin3d.py:333-342); with ss == 0 the last slice is `slice(0, None)`, i.e. it relabels the
    whole axis with 2.
    """
    if ss == 0:
        return np.full(n, 2, dtype=np.int64)
    x = np.arange(n)
    return np.where(x < n - ws, 0, np.where(x < n - ss, 1, 2)).astype(np.int64)


@lru_cache(maxsize=64)
def window_tables(res: Res, window: Res, shifted: bool) -> tuple[np.ndarray, np.ndarray | None, Res]:
    """(tok, grp or None, effective window size) for one block flavour of one stage."""
    C, H, W = res
    base_shift = tuple(w // 2 for w in window) if shifted else (0, 0, 0)
    ws, ss = adjust_windows(window, base_shift, res)
    dims = (C, H, W)
    pad = tuple((-d) % w for d, w in zip(dims, ws))
    front = tuple(p // 2 for p in pad)  # two-sided padding, front = pad // 2 (swin3d.py:177-194)
    padded = tuple(d + p for d, p in zip(dims, pad))

    # Coordinates of every padded, rolled position, in window-partition order
    # (c1, h1, w1 | wc, wh, ww)  (swin3d.py:212-213).
    axes = []
    for a in range(3):
        n_win = padded[a] // ws[a]
        axes.append((np.arange(n_win)[:, None] * ws[a] + np.arange(ws[a])[None, :]))  # (n_win, ws)
    pc = axes[0][:, None, None, :, None, None]
    ph = axes[1][None, :, None, None, :, None]
    pw = axes[2][None, None, :, None, None, :]
    shape = (axes[0].shape[0], axes[1].shape[0], axes[2].shape[0], ws[0], ws[1], ws[2])
    pc, ph, pw = (np.broadcast_to(a, shape) for a in (pc, ph, pw))

    rolled = [pc - front[0], ph - front[1], pw - front[2]]  # position in the rolled, unpadded grid
    valid = np.ones(shape, dtype=bool)
    for a in range(3):
        valid &= (rolled[a] >= 0) & (rolled[a] < dims[a])
    # torch.roll(x, -s): rolled[i] = x[(i + s) % n]
    orig = [(rolled[a] + ss[a]) % dims[a] for a in range(3)]
    tok = (orig[0] * H + orig[1]) * W + orig[2]
    tok = np.where(valid, tok, -1).astype(np.int32)

    n_windows = shape[0] * shape[1] * shape[2]
    n_tok = ws[0] * ws[1] * ws[2]
    tok = np.ascontiguousarray(tok.reshape(n_windows, n_tok))

    grp = None
    if any(ss):
        lab = [_axis_labels(dims[a], ws[a], ss[a]) for a in range(3)]
        g = np.zeros(shape, dtype=np.int64)
        for a, r in enumerate(rolled):
            la = lab[a][np.clip(r, 0, dims[a] - 1)]
            if a == 2:
                la = np.where(la == 1, 2, la)  # longitude wraps: W slices 1 and 2 communicate
            g = g * 3 + la
        g = np.where(valid, g, 27)  # padding is its own group
        grp = np.ascontiguousarray(g.reshape(n_windows, n_tok).astype(np.uint8))
    return tok, grp, ws
