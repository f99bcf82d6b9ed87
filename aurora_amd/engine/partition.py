"""Latitude-band partition of the token grid and the halo plans of window attention.

Strong scaling of ONE forecast over R GPUs: every rank owns a contiguous band
of latitude rows at every backbone stage.  All operators except window attention are local to a
token (GEMMs, LayerNorms, MLPs), to a 2x2 block (patch merge/split: band boundaries are kept on even
rows of the finer stage) or to a grid column (patch embed, Perceiver level (de)aggregation,
unpatchify), so they simply run on the rank's rows.

Window attention needs, for every window that contains at least one owned token, the q/k/v rows of
the window's other tokens.  Because the engine's attention kernel already gathers through a token
table, a band needs no special kernel: its table indexes a local buffer `[own rows | halo rows]`,
the halo rows are received from the neighbouring ranks, and outputs are written for owned tokens
only.  Windows that straddle a boundary are evaluated on both sides (each for its own queries).

Foreign tokens that no owned query can see are not exchanged at all: in a shifted block the -100 mask
(swin3d.py:333-358) only lets tokens of the same group attend to each other, so a foreign position whose group
contains no owned position of that window is replaced by an absent one (token -1, its group label kept: it stays
masked, with weight exp(-100) ~ 4e-44 instead of a real key's exp(-100 + ...)).  This is what makes the cyclic wrap
of the latitude roll free: the window row that holds the last three and -- rolled -- the first
three latitude rows puts them in different groups, so the first and the last rank never talk to each other.

Everything here is host-side numpy, derived from `geometry.window_tables` (which is itself checked
against the reference's roll/pad/partition chain).  tests/test_partition.py replays the plans with
numpy attention -- in-process for several rank counts and across two gloo processes.
"""

from __future__ import annotations

import dataclasses
import math
from functools import lru_cache

import numpy as np

from aurora_amd.engine import geometry

Res = tuple[int, int, int]


def _rows_from_bounds(all_res: list[Res], bounds: list[int]) -> list[list[tuple[int, int]]] | None:
    """Owned rows `[stage][rank]` from boundaries on the coarsest stage; None if a rank ends up without rows."""
    n, world = len(all_res), len(bounds) - 1
    out = []
    for s in range(n):
        mult, Hs = 2 ** (n - 1 - s), all_res[s][1]
        rows = [(min(bounds[r] * mult, Hs), min(bounds[r + 1] * mult, Hs) if r < world - 1 else Hs) for r in range(world)]
        if any(h1 <= h0 for h0, h1 in rows):
            return None
        out.append(rows)
    return out


SEARCH_BUDGET = 200_000   # candidate partitions tried before giving up (csrc/band.h: BAND_SEARCH_BUDGET)


def _group_spans(all_res: list[Res], window: Res) -> list[np.ndarray]:
    """Per stage, the distinct (lowest, highest) latitude rows of the sets of tokens that attend to each other: a window's
    positions of one mask group, in both block flavours (only the latitude structure matters: a grid one window wide).
    They depend on the grid alone -- collected once per search, not per candidate (csrc/band.hip: group_spans)."""
    out = []
    for C, H, W in all_res:
        res = (C, H, min(W, window[2]))
        spans = set()
        for shifted in (False, True):
            tok, grp, _ = geometry.window_tables(res, window, shifted)
            row = np.where(tok >= 0, (np.maximum(tok, 0) // res[2]) % H, -1)
            g = grp if grp is not None else np.zeros_like(tok, dtype=np.uint8)
            for label in np.unique(g):
                sel = (g == label) & (tok >= 0)
                hi = np.where(sel, row, -1).max(axis=1)
                lo = np.where(sel, row, 1 << 30).min(axis=1)
                spans.update((int(a), int(b)) for a, b in zip(lo, hi) if b > a)
        out.append(np.array(sorted(spans), dtype=np.int64).reshape(-1, 2))
    return out


def _neighbours_suffice(spans: list[np.ndarray], all_res: list[Res], rows: list[list[tuple[int, int]]]) -> bool:
    """Tokens that attend to each other lie on at most two adjacent ranks at every stage: bands are contiguous and ordered,
    so a set of rows spans the ranks owner[lowest row] .. owner[highest row]."""
    for s, (C, H, W) in enumerate(all_res):
        owner = np.full(H, -1)
        for r, (h0, h1) in enumerate(rows[s]):
            owner[h0:h1] = r
        sp = spans[s]
        if len(sp) and np.any(owner[sp[:, 1]] - owner[sp[:, 0]] > 1):
            return False
    return True


def _compositions(world: int, total: int, m: int):
    """`world` band sizes in 1 .. m summing to `total`, thick bands first (the order csrc/band.hip searches in)."""
    if world == 1:
        if 1 <= total <= m:
            yield [total]
        return
    for sz in range(min(m, total - (world - 1)), max(1, total - (world - 1) * m) - 1, -1):
        for rest in _compositions(world - 1, total - sz, m):
            yield [sz] + rest


def band_rows(all_res: list[Res], window: Res, world: int) -> list[list[tuple[int, int]]]:
    """Owned latitude rows `[stage][rank] -> (h0, h1)`.

    Boundaries are chosen on the coarsest stage and doubled per finer stage, so 2x2 merges / splits
    never cross a rank.  First choice: a unit such that the finer stages' boundaries fall on window rows
    (then un-shifted blocks need no halo there).  When that split is badly balanced, or bands get so thin that a
    window would reach past a whole band (halo rows from a rank that is not a neighbour), the partitions with the
    smallest largest band are searched, thick bands first, for one whose windows stay within neighbouring ranks.
    """
    n = len(all_res)
    Hc = all_res[-1][1]
    if Hc < world:
        raise ValueError(f"cannot split {Hc} latitude rows of the coarsest stage over {world} ranks")
    unit = window[1] // math.gcd(window[1], 2) if n > 1 else window[1]
    if unit < 1 or -(-Hc // unit) < world:
        unit = 1
    n_units = -(-Hc // unit)
    base, extra = divmod(n_units, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + (base + (1 if r < extra else 0)) * unit)
    bounds = [min(b, Hc) for b in bounds]
    bounds[-1] = Hc
    rows = _rows_from_bounds(all_res, bounds)
    m_opt, m_unit = -(-Hc // world), max(b1 - b0 for b0, b1 in zip(bounds, bounds[1:]))
    # kept unless badly balanced (largest band more than 1/12 above the smallest possible largest band)
    spans = _group_spans(all_res, window)
    if rows is not None and (m_unit - m_opt) * 12 <= m_opt and _neighbours_suffice(spans, all_res, rows):
        return rows
    budget = SEARCH_BUDGET
    for m in range(-(-Hc // world), Hc + 1):
        for sizes in _compositions(world, Hc, m):
            budget -= 1
            if budget < 0:
                raise ValueError(f"none of the first {SEARCH_BUDGET} splits of {Hc} coarsest-stage rows over {world} ranks (most "
                                 "balanced first) keeps every window within two neighbouring ranks; search stopped (bands too "
                                 "thin: use fewer ranks)")
            rows = _rows_from_bounds(all_res, [0] + [int(x) for x in np.cumsum(sizes)])
            if rows is not None and _neighbours_suffice(spans, all_res, rows):
                return rows
    raise ValueError(f"no split of {Hc} coarsest-stage rows over {world} ranks keeps every window within two neighbouring "
                     "ranks (bands too thin)")


@dataclasses.dataclass
class BlockPlan:
    """What rank `rank` needs to run window attention of one block flavour on its band."""

    tok: np.ndarray                  # int32 [n_windows_local, N]: index into [own | halo], -1 = padding
    grp: np.ndarray | None           # uint8 [n_windows_local, N] or None
    n_own: int                       # owned tokens (rows of the local activation buffers)
    n_halo: int                      # halo rows appended behind them
    recv: dict[int, tuple[int, int]]  # peer -> (offset into the halo region, count)
    send: dict[int, np.ndarray]      # peer -> int32 local indices of owned tokens, in the peer's halo order


def local_index(tokens: np.ndarray, res: Res, h0: int, h1: int) -> np.ndarray:
    """Local index (c * rows + h - h0) * W + w of global token ids that lie in rows [h0, h1)."""
    C, H, W = res
    c, rem = np.divmod(tokens, H * W)
    h, w = np.divmod(rem, W)
    return ((c * (h1 - h0) + (h - h0)) * W + w).astype(np.int64)


@lru_cache(maxsize=256)
def block_plans(res: Res, window: Res, shifted: bool, rows: tuple[tuple[int, int], ...]) -> tuple[BlockPlan, ...]:
    """Plans of all ranks for one (stage resolution, block flavour); `rows[rank] = (h0, h1)`."""
    C, H, W = res
    world = len(rows)
    tok_g, grp_g, _ = geometry.window_tables(res, window, shifted)
    real = tok_g >= 0
    h_of = (np.where(real, tok_g, 0) // W) % H
    owner_of_row = np.empty(H, dtype=np.int64)
    for r, (h0, h1) in enumerate(rows):
        owner_of_row[h0:h1] = r
    owner = np.where(real, owner_of_row[h_of], -1)  # [nW, N]

    foreign_lists = []
    plans = []
    for r, (h0, h1) in enumerate(rows):
        mine = (owner == r).any(axis=1)                      # windows touching the band
        tw, ow = tok_g[mine].copy(), owner[mine].copy()
        gw = None if grp_g is None else np.ascontiguousarray(grp_g[mine])
        if gw is not None:
            # drop the foreign positions whose mask group holds no owned position of the same window
            bits = np.bitwise_or.reduce(np.where(ow == r, np.uint32(1) << gw.astype(np.uint32), np.uint32(0)), axis=1)
            visible = ((bits[:, None] >> gw.astype(np.uint32)) & 1).astype(bool)
            unseen = (ow != r) & (ow >= 0) & ~visible
            tw[unseen] = -1
            ow[unseen] = -1
        foreign = np.unique(tw[(ow != r) & (ow >= 0)])       # sorted global ids
        f_owner = owner_of_row[(foreign // W) % H]
        order = np.lexsort((foreign, f_owner))               # group by owning rank, then by id
        foreign, f_owner = foreign[order], f_owner[order]
        n_own = C * (h1 - h0) * W
        halo_pos = {int(t): n_own + i for i, t in enumerate(foreign)}
        loc = np.full(tw.shape, -1, dtype=np.int64)
        own_mask = ow == r
        loc[own_mask] = local_index(tw[own_mask], res, h0, h1)
        fm = (ow != r) & (ow >= 0)
        if fm.any():
            loc[fm] = np.vectorize(halo_pos.__getitem__, otypes=[np.int64])(tw[fm])
        recv = {}
        for q in np.unique(f_owner):
            idx = np.nonzero(f_owner == q)[0]
            recv[int(q)] = (int(idx[0]), int(len(idx)))
        foreign_lists.append((foreign, f_owner))
        plans.append(BlockPlan(tok=np.ascontiguousarray(loc.astype(np.int32)), grp=gw, n_own=n_own,
                               n_halo=len(foreign), recv=recv, send={}))
    # what I send = what my peers listed as foreign and I own, in THEIR order
    for q, (foreign, f_owner) in enumerate(foreign_lists):
        for r in np.unique(f_owner):
            toks = foreign[f_owner == r]
            h0, h1 = rows[int(r)]
            plans[int(r)].send[q] = local_index(toks, res, h0, h1).astype(np.int32)
    for r, p in enumerate(plans):
        assert set(p.send) == {q for q, pq in enumerate(plans) if r in pq.recv}
    return tuple(plans)
