"""Latitude-band plans: replaying them (numpy attention on `[own | halo]` buffers, halos moved exactly as
the send/recv lists say) reproduces global window attention; once in-process for several rank counts and
once across two gloo processes."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from aurora_amd.engine import geometry, partition

WINDOW = (2, 6, 12)


def attention_numpy(buf, tok, grp, D, heads):
    """Per-window softmax attention on rows of `buf` ([n, 3D], q|k|v); returns {local row: output}."""
    hd = D // heads
    out = {}
    for w in range(tok.shape[0]):
        idx = tok[w]
        rows = np.where(idx[:, None] >= 0, buf[np.maximum(idx, 0)], 0.0)  # padding: q = k = v = bias = 0 here
        q, k, v = (rows[:, i * D:(i + 1) * D].reshape(-1, heads, hd).transpose(1, 0, 2) for i in range(3))
        s = q @ k.transpose(0, 2, 1) / np.sqrt(hd)
        if grp is not None:
            s = s + np.where(grp[w][None, :, None] != grp[w][None, None, :], -100.0, 0.0)
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        o = (p / p.sum(-1, keepdims=True)) @ v
        o = o.transpose(1, 0, 2).reshape(-1, D)
        for n, t in enumerate(idx):
            if t >= 0:
                out[int(t)] = o[n]
    return out


def global_attention(qkv, res, shifted, D, heads):
    tok, grp, _ = geometry.window_tables(res, WINDOW, shifted)
    o = attention_numpy(qkv, tok, grp, D, heads)
    return np.stack([o[t] for t in range(qkv.shape[0])])


def rank_inputs(qkv, res, rows_r):
    C, H, W = res
    h0, h1 = rows_r
    return qkv.reshape(C, H, W, -1)[:, h0:h1].reshape(-1, qkv.shape[1])


CASES = [((4, 24, 24), 2), ((4, 24, 24), 4), ((4, 45, 30), 3), ((4, 45, 24), 8), ((4, 13, 26), 2)]


@pytest.mark.parametrize("res,world", CASES)
@pytest.mark.parametrize("shifted", [False, True])
def test_band_plans_reproduce_global_attention(res, world, shifted):
    C, H, W = res
    D, heads = 8, 2
    rng = np.random.default_rng(0)
    qkv = rng.standard_normal((C * H * W, 3 * D))
    ref = global_attention(qkv, res, shifted, D, heads)
    rows = partition.band_rows([res], WINDOW, world)[0]
    assert rows[0][0] == 0 and rows[-1][1] == H and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    plans = partition.block_plans(res, WINDOW, shifted, tuple(rows))
    own = [rank_inputs(qkv, res, rows[r]) for r in range(world)]
    bufs = []
    for r, p in enumerate(plans):
        assert own[r].shape[0] == p.n_own
        halo = np.zeros((p.n_halo, 3 * D))
        for q, (off, cnt) in p.recv.items():
            sent = own[q][plans[q].send[r]]            # what peer q sends to r
            assert sent.shape[0] == cnt
            halo[off:off + cnt] = sent
        bufs.append(np.concatenate([own[r], halo]))
    got = np.zeros_like(ref)
    seen = np.zeros(C * H * W, dtype=int)
    for r, p in enumerate(plans):
        out = attention_numpy(bufs[r], p.tok, p.grp, D, heads)
        h0, h1 = rows[r]
        for loc, val in out.items():
            if loc < p.n_own:                           # outputs only for owned tokens
                c, rem = divmod(loc, (h1 - h0) * W)
                hl, w_ = divmod(rem, W)
                g = (c * H + h0 + hl) * W + w_
                got[g] = val
                seen[g] += 1
    assert (seen == 1).all()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12)


def test_band_rows_are_merge_aligned_and_window_aligned_where_possible():
    all_res, _ = geometry.stage_resolutions((4, 180, 360), 3)
    rows = partition.band_rows(all_res, WINDOW, 8)
    for r in range(8):
        (a0, b0), (a1, b1), (a2, b2) = rows[0][r], rows[1][r], rows[2][r]
        assert (a0, a1) == (4 * a2, 2 * a2) and b0 == min(4 * b2, 180) and b1 == min(2 * b2, 90)
        assert a0 % 6 == 0 and a1 % 6 == 0          # window-aligned at stages 0 and 1
    sizes = [b - a for a, b in rows[2]]
    assert sum(sizes) == 45 and max(sizes) == 6
    # un-shifted blocks need no halo where bands are window-aligned
    for s in (0, 1):
        for p in partition.block_plans(all_res[s], WINDOW, False, tuple(rows[s])):
            assert p.n_halo == 0 and not p.send
        # shifted blocks: 3 halo rows per neighbouring side.  The cyclic wrap of the latitude roll costs nothing: the
        # mask keeps the wrapped rows apart, so the first and the last rank have ONE neighbour each
        C, _, W = all_res[s]
        plans = partition.block_plans(all_res[s], WINDOW, True, tuple(rows[s]))
        for r, p in enumerate(plans):
            sides = 1 if r in (0, len(plans) - 1) else 2
            assert p.n_halo == sides * 3 * C * W and len(p.send) == sides and len(p.recv) == sides
            assert all(abs(q - r) == 1 for q in list(p.send) + list(p.recv))


def _gloo_worker(rank, world, port, res, shifted, ret):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C, H, W = res
    D, heads = 8, 2
    qkv = np.random.default_rng(0).standard_normal((C * H * W, 3 * D))
    rows = partition.band_rows([res], WINDOW, world)[0]
    p = partition.block_plans(res, WINDOW, shifted, tuple(rows))[rank]
    own = torch.from_numpy(rank_inputs(qkv, res, rows[rank]).copy())
    halo = torch.zeros((p.n_halo, 3 * D), dtype=torch.float64)
    ops, keep = [], []
    for q, idx in p.send.items():
        t = own[torch.from_numpy(idx.astype(np.int64))].contiguous()
        keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, q))
    for q, (off, cnt) in p.recv.items():
        ops.append(dist.P2POp(dist.irecv, halo[off:off + cnt], q))
    for req in (dist.batch_isend_irecv(ops) if ops else []):   # (a split on a window row needs no exchange when un-shifted)
        req.wait()
    out = attention_numpy(torch.cat([own, halo]).numpy(), p.tok, p.grp, D, heads)
    band = np.stack([out[i] for i in range(p.n_own)])
    gathered = [None] * world
    dist.all_gather_object(gathered, band)
    if rank == 0:
        full = np.concatenate([g.reshape(C, -1, W, D) for g in gathered], axis=1).reshape(-1, D)
        ref = global_attention(qkv, res, shifted, D, heads)
        ret.put(float(np.abs(full - ref).max()))
    dist.destroy_process_group()


@pytest.mark.parametrize("shifted", [False, True])
def test_halo_exchange_over_two_gloo_processes(shifted):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29650 + int(shifted)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, (4, 22, 24), shifted, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ret.get(timeout=10) < 1e-10


def test_partition_search_is_bounded_and_says_when_it_gave_up():
    """A rank count the grid cannot carry must fail at once, not after minutes (the check of a candidate costs a few hundred
    look-ups since the row spans of the mask groups are collected once per grid; both twins stop after the same number of
    candidates and say so), and the 12- / 16-way splits of the 0.25-degree token grid, which the slow search of round 3 ran
    out of budget on, are found -- with plans that need neighbours only."""
    import ctypes
    import time

    from aurora_amd.engine import lib

    L = lib.load()
    i32 = lambda v: (ctypes.c_int32 * len(v))(*v)  # noqa: E731
    h0, h1 = ctypes.c_int32(), ctypes.c_int32()
    t0 = time.perf_counter()
    rc = L.aurora_hip_band_partition(3, i32((4, 180, 360)), i32(WINDOW), 24, 0, 0, ctypes.byref(h0), ctypes.byref(h1))
    assert rc != 0 and time.perf_counter() - t0 < 5.0
    assert b"search stopped" in L.aurora_hip_last_error() and str(partition.SEARCH_BUDGET).encode() in L.aurora_hip_last_error()
    all_res = [(4, 180, 360), (4, 90, 180), (4, 45, 90)]
    for world in (12, 16):
        rows = partition.band_rows(all_res, WINDOW, world)
        assert _c_rows(3, (4, 180, 360), WINDOW, world) == [[tuple(r) for r in st] for st in rows]
        for s, res in enumerate(all_res):
            for shifted in (False, True):
                partition.block_plans(res, WINDOW, shifted, tuple(rows[s]))   # raises if a halo row is not a neighbour's


def _selftest_worker(rank, world, port, corrupt, ret):
    import os

    import torch.distributed as dist

    from aurora_amd.engine import native
    from aurora_amd.engine.engine import Shard

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = native._Transport(Shard(rank, world, None), "cpu")
    if corrupt and rank == 1:   # a transport that delivers the wrong bytes must be caught by the receiver
        good = t._wait

        def bad(user, stream):
            rc = good(user, stream)
            t.recv[5] ^= 1
            return rc

        t._wait = bad
    try:
        t.selftest(4096)
        ret.put((rank, "ok", t.exchanges))
    except RuntimeError as e:
        ret.put((rank, str(e), t.exchanges))
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_transport_selftest_over_three_gloo_processes(corrupt):
    """What bench.py --gpus N runs before it times anything: every rank sends a rank-stamped pattern to its neighbours through
    the production `_Transport` (its staging buffers, `_post` / `_wait`, the process group) and verifies what arrived."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_selftest_worker, args=(r, 3, 29670 + int(corrupt), corrupt, ret)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = dict((r, (msg, n)) for r, msg, n in (ret.get(timeout=10) for _ in range(3)))
    assert all(n == 1 for _, n in got.values())
    if corrupt:
        assert got[0][0] == got[2][0] == "ok" and "wrong byte 5 from rank 0" in got[1][0]
    else:
        assert all(msg == "ok" for msg, _ in got.values())


@pytest.mark.parametrize("res,world", [((4, 24, 24), 2), ((4, 45, 24), 8), ((4, 180, 360), 8)])
@pytest.mark.parametrize("shifted", [False, True])
def test_interior_windows_need_no_halo_and_cover_most_of_a_band(res, world, shifted):
    """The engine attends the windows that reference no halo row while the exchange is in flight
    (`Engine._plans`: interior / boundary).  Interior windows must index owned rows only, the two classes
    partition the plan, and at the benchmark geometry at least half of the windows are interior."""
    rows = partition.band_rows([res], WINDOW, world)[0]
    plans = partition.block_plans(res, WINDOW, shifted, tuple(rows))
    for p in plans:
        needs_halo = (p.tok >= p.n_own).any(axis=1)
        interior, boundary = p.tok[~needs_halo], p.tok[needs_halo]
        assert (interior < p.n_own).all()
        assert len(interior) + len(boundary) == len(p.tok)
        if not (p.recv or p.send):
            assert not needs_halo.any()
        if res == (4, 180, 360) and p.n_halo:
            assert len(interior) >= 0.5 * len(p.tok)   # 8 ranks: 3 (or 2) of the 5 (4) shifted window rows per band


# ---- the C++ twins inside libaurora_hip.so (csrc/band.hip): what the model handle really uses -----------------------
def _c_rows(n_stages, res0, window, world):
    import ctypes

    from aurora_amd.engine import lib

    L = lib.load()
    i32 = lambda v: (ctypes.c_int32 * len(v))(*v)  # noqa: E731
    out = []
    for s in range(n_stages):
        rows = []
        for r in range(world):
            h0, h1 = ctypes.c_int32(), ctypes.c_int32()
            assert L.aurora_hip_band_partition(n_stages, i32(res0), i32(window), world, r, s, ctypes.byref(h0),
                                               ctypes.byref(h1)) == 0, L.aurora_hip_last_error()
            rows.append((h0.value, h1.value))
        out.append(rows)
    return out


def _c_plan(res, window, shifted, world, rank, rows):
    import ctypes

    from aurora_amd.engine import lib

    L = lib.load()
    i32 = lambda v: (ctypes.c_int32 * len(v))(*v)  # noqa: E731
    flat = i32([x for r in rows for x in r])
    info = lib.HipPlanInfo()
    args = (i32(res), i32(window), int(shifted), world, rank, flat, ctypes.byref(info))
    assert L.aurora_hip_band_plan(*args, None, None, None, None) == 0, L.aurora_hip_last_error()
    tok = np.empty((info.n_windows, info.win_tokens), np.int32)
    grp = np.empty((info.n_windows, info.win_tokens), np.uint8)
    sp, sn = np.empty(info.send_count[0], np.int32), np.empty(info.send_count[1], np.int32)
    assert L.aurora_hip_band_plan(*args, tok.ctypes.data, grp.ctypes.data, sp.ctypes.data, sn.ctypes.data) == 0
    return info, tok, (grp if info.has_groups else None), sp, sn


@pytest.mark.parametrize("res0,window,n_stages,world", [
    ((4, 180, 360), (2, 6, 12), 3, 8), ((4, 180, 360), (2, 6, 12), 3, 2), ((4, 150, 300), (2, 6, 12), 3, 4),
    ((4, 45, 90), (2, 6, 12), 3, 3), ((4, 12, 8), (2, 4, 4), 2, 3), ((2, 9, 6), (2, 3, 3), 1, 3),
    # thin bands: the window-aligned split would need halo rows from a rank that is not a neighbour -> searched split
    ((4, 150, 300), (2, 6, 12), 3, 8), ((4, 64, 24), (2, 6, 12), 3, 5), ((4, 56, 24), (2, 6, 12), 3, 6),
])
def test_c_partition_and_plans_equal_the_numpy_ones(res0, window, n_stages, world):
    """csrc/band.hip (aurora_hip_band_partition / aurora_hip_band_plan) against partition.py, whose plans the tests above
    replay against global attention: same owned rows; per rank the same windows, token / group tables (the C++ plan
    lists the windows that touch no halo row first), receive slices and send lists."""
    all_res, _ = geometry.stage_resolutions(res0, n_stages)
    want_rows = partition.band_rows(all_res, window, world)
    assert _c_rows(n_stages, res0, window, world) == [[tuple(r) for r in rows] for rows in want_rows]
    for stage, res in enumerate(all_res):
        for shifted in (False, True):
            plans = partition.block_plans(tuple(res), tuple(window), shifted, tuple(want_rows[stage]))
            for rank, p in enumerate(plans):
                info, tok, grp, sp, sn = _c_plan(res, window, shifted, world, rank, want_rows[stage])
                assert (info.n_own, info.n_halo, info.n_windows) == (p.n_own, p.n_halo, p.tok.shape[0])
                halo = (p.tok >= p.n_own).any(axis=1)
                order = np.concatenate([np.nonzero(~halo)[0], np.nonzero(halo)[0]])
                assert info.n_interior == int((~halo).sum())
                assert np.array_equal(tok, p.tok[order])
                assert (grp is None) == (p.grp is None)
                if grp is not None:
                    assert np.array_equal(grp, p.grp[order])
                for side, q, sent in ((0, rank - 1, sp), (1, rank + 1, sn)):
                    if q in p.recv:
                        assert (info.recv_offset[side], info.recv_count[side]) == p.recv[q]
                    else:
                        assert info.recv_count[side] == 0
                    assert np.array_equal(sent, p.send.get(q, np.empty(0, np.int32)))
                assert set(p.recv) <= {rank - 1, rank + 1} and set(p.send) <= {rank - 1, rank + 1}


def test_thin_bands_get_a_searched_partition():
    """The 0.4-degree grid on 8 ranks (BASELINE configs[4]): 38 coarsest-stage rows; the window-aligned split 6 x 5 + 3 + 3 + 2
    lets a 6-row window reach past a 3-row band.  The search finds 7 x 5 + 3: every window within two neighbouring ranks,
    largest band 5 of 38 rows (load-balance bound 0.95)."""
    all_res, _ = geometry.stage_resolutions((4, 150, 300), 3)
    rows = partition.band_rows(all_res, WINDOW, 8)
    assert [h1 - h0 for h0, h1 in rows[-1]] == [5, 5, 5, 5, 5, 5, 5, 3]
    assert [h1 - h0 for h0, h1 in rows[0]] == [20, 20, 20, 20, 20, 20, 20, 10]
    for s, res in enumerate(all_res):
        for shifted in (False, True):
            for rank, p in enumerate(partition.block_plans(res, WINDOW, shifted, tuple(rows[s]))):
                assert set(p.recv) <= {rank - 1, rank + 1} and set(p.send) <= {rank - 1, rank + 1}
    # the grids whose window-aligned split works keep it (nothing validated so far changes)
    era5, _ = geometry.stage_resolutions((4, 180, 360), 3)
    assert [h1 - h0 for h0, h1 in partition.band_rows(era5, WINDOW, 8)[0]] == [24] * 7 + [12]
    with pytest.raises(ValueError, match="too thin"):
        partition.band_rows(geometry.stage_resolutions((4, 24, 24), 3)[0], WINDOW, 3)


def test_c_partition_says_why_it_cannot_split():
    import ctypes

    from aurora_amd.engine import lib

    L = lib.load()
    i32 = lambda v: (ctypes.c_int32 * len(v))(*v)  # noqa: E731
    h0, h1 = ctypes.c_int32(), ctypes.c_int32()
    assert L.aurora_hip_band_partition(3, i32((4, 8, 16)), i32((2, 6, 12)), 8, 0, 0, ctypes.byref(h0), ctypes.byref(h1)) == -1
    assert b"cannot split" in L.aurora_hip_last_error()


def test_partition_search_properties():
    """Random grids / rank counts: whenever a partition exists, csrc/band.hip and partition.py agree on it, the bands tile
    the rows of every stage in order, 2 x 2 merges never straddle a rank, and every rank's halo rows come from its two
    neighbours (in both block flavours at every stage)."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    @settings(max_examples=120, deadline=None)
    @given(st.integers(6, 70), st.integers(2, 8), st.integers(1, 3), st.sampled_from([(2, 6, 12), (2, 4, 4), (1, 3, 6)]))
    def check(h0, world, n_stages, window):
        res0 = (4, 2 * h0, 12)
        all_res, _ = geometry.stage_resolutions(res0, n_stages)
        try:
            want = partition.band_rows(all_res, window, world)
        except ValueError:
            h_a, h_b = ctypes.c_int32(), ctypes.c_int32()
            rc = L.aurora_hip_band_partition(n_stages, i32(res0), i32(window), world, 0, 0, ctypes.byref(h_a), ctypes.byref(h_b))
            assert rc == -1     # the library refuses the same cases
            return
        assert _c_rows(n_stages, res0, window, world) == [[tuple(r) for r in rows] for rows in want]
        for s, res in enumerate(all_res):
            rows = want[s]
            assert rows[0][0] == 0 and rows[-1][1] == res[1] and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            if s + 1 < len(all_res):
                assert all(h1 % 2 == 0 for _, h1 in rows[:-1])    # a merge pair (2k, 2k + 1) belongs to one rank
            for shifted in (False, True):
                for rank, p in enumerate(partition.block_plans(tuple(res), tuple(window), shifted, tuple(rows))):
                    assert set(p.recv) <= {rank - 1, rank + 1} and set(p.send) <= {rank - 1, rank + 1}

    import ctypes

    from aurora_amd.engine import lib

    L = lib.load()
    i32 = lambda v: (ctypes.c_int32 * len(v))(*v)  # noqa: E731
    check()
