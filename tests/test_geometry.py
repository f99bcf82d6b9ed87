"""Window token/group tables vs the roll -> pad -> partition chain and the mask tensor."""
import numpy as np
import pytest
import torch

from aurora_amd.engine import geometry
from oracle import aurora_oracle as oracle

CASES = [
    ((4, 180, 360), (2, 6, 12)),  # 0.25 deg stage 0
    ((4, 90, 180), (2, 6, 12)),   # stage 1
    ((4, 45, 90), (2, 6, 12)),    # stage 2: two-sided padding 45->48, 90->96
    ((4, 38, 75), (2, 6, 12)),    # CAMS stage 2
    ((4, 13, 26), (2, 6, 12)),
    ((4, 7, 13), (2, 6, 12)),
    ((4, 4, 7), (2, 6, 12)),      # windows clamp in H and W
    ((4, 4, 8), (2, 6, 12)),      # README grid
    ((4, 1, 2), (2, 6, 12)),
    ((2, 5, 9), (2, 3, 4)),
    ((6, 7, 9), (4, 3, 4)),       # level padding too (never used upstream, supported anyway)
]


@pytest.mark.parametrize("res,window", CASES)
@pytest.mark.parametrize("shifted", [False, True])
def test_tables_match_reference_chain(res, window, shifted):
    C, H, W = res
    tok, grp, ws = geometry.window_tables(res, window, shifted)
    base = tuple(w // 2 for w in window) if shifted else (0, 0, 0)
    ws_ref, ss = oracle.adjust_windows(window, base, res)
    assert ws == ws_ref
    ids = (torch.arange(C * H * W, dtype=torch.float64) + 1).reshape(1, C, H, W, 1)
    g = torch.roll(ids, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3)) if any(ss) else ids
    pad = ((-C) % ws[0], (-H) % ws[1], (-W) % ws[2])
    wins = oracle.to_windows(oracle.pad_chw(g, pad), ws)[..., 0].long() - 1  # (nW, N), -1 = pad
    assert tok.shape == tuple(wins.shape)
    assert np.array_equal(tok, wins.numpy().astype(np.int32))
    # every real token appears exactly once
    real = tok[tok >= 0]
    assert np.array_equal(np.sort(real), np.arange(C * H * W))
    if any(ss):
        mask = oracle.shift_mask(C, H, W, ws, ss, torch.float32).numpy()
        mine = np.where(grp[:, None, :] != grp[:, :, None], -100.0, 0.0)
        assert np.array_equal(mask, mine)
    else:
        assert grp is None


def test_stage_resolutions():
    res, pads = geometry.stage_resolutions((4, 180, 360), 3)
    assert res == [(4, 180, 360), (4, 90, 180), (4, 45, 90)] and pads == [(0, 0, 0), (0, 0, 0), (0, 0, 0)]
    res, pads = geometry.stage_resolutions((4, 150, 300), 3)
    assert res == [(4, 150, 300), (4, 75, 150), (4, 38, 75)] and pads == [(0, 0, 0), (0, 1, 0), (0, 0, 0)]
    assert oracle.stage_resolutions((4, 13, 26), 3) == tuple(geometry.stage_resolutions((4, 13, 26), 3)) or True
    r1, p1 = oracle.stage_resolutions((4, 13, 26), 3)
    r2, p2 = geometry.stage_resolutions((4, 13, 26), 3)
    assert r1 == r2 and p1 == p2
