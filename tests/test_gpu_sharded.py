"""Latitude-band sharding on ONE GPU: R engines of the same model run as virtual ranks in one process;
their halo exchanges are carried out by copying exactly the tensors the plans name (the RCCL
point-to-point path moves the same tensors between processes).  The stitched bands must equal the
un-sharded engine."""
import dataclasses
from collections import defaultdict, deque

import pytest
import torch

import aurora_amd
from aurora_amd import Batch, Metadata
from aurora_amd.batch import BandBatch
from aurora_amd.engine.engine import Complete, Engine, Shard
from tests import helpers
from tests.golden_cases import CASES

pytestmark = pytest.mark.gpu


def make_engines(model, world):
    engines = []
    for r in range(world):
        model._shard = Shard(r, world, None, gather_output=False)
        engines.append(Engine(model))
    model._shard = None
    return engines


def run_virtual_ranks(model, batch, world, engines=None):
    """Advance `world` sharded step generators until all finish; returns the per-rank BandBatches."""
    engines = engines or make_engines(model, world)
    gens = [e.step_gen(batch if not isinstance(batch, list) else batch[r]) for r, e in enumerate(engines)]
    mailbox = defaultdict(deque)           # (src, dst) -> tensors in posting order
    waiting, done = {}, {}
    n_exchanges = 0

    def advance(r, first=False):
        try:
            req = next(gens[r]) if first else gens[r].send(None)
            while isinstance(req, Complete):   # (the harness completes every exchange before resuming a rank)
                req = gens[r].send(None)
        except StopIteration as fin:
            done[r] = fin.value
            return
        for peer, t in req.sends:
            mailbox[(r, peer)].append(t)
        waiting[r] = req

    for r in range(world):
        advance(r, first=True)
    while len(done) < world:
        progressed = False
        for r in range(world):
            req = waiting.get(r)
            if req is None:
                continue
            need = defaultdict(int)
            for peer, _ in req.recvs:
                need[peer] += 1
            if all(len(mailbox[(peer, r)]) >= n for peer, n in need.items()):
                for peer, dst in req.recvs:
                    src = mailbox[(peer, r)].popleft()
                    assert src.shape == dst.shape and src.dtype == dst.dtype
                    dst.copy_(src)
                del waiting[r]
                n_exchanges += 1
                advance(r)
                progressed = True
        assert progressed, "virtual ranks deadlocked"
    assert all(len(q) == 0 for q in mailbox.values())
    return [done[r] for r in range(world)], n_exchanges


def stitch(bands):
    cat = lambda key: {k: torch.cat([getattr(b, key)[k] for b in bands], dim=-2) for k in getattr(bands[0], key)}  # noqa: E731
    return cat("surf_vars"), cat("atmos_vars")


def make(name, autocast, H=None, W=None):
    case = dict(CASES[name])
    if H:
        case["H"], case["W"] = H, W
    model = getattr(aurora_amd, case["cls"])(**case["kwargs"], autocast=autocast)
    model.load_state_dict(helpers.case_state_dict(model, torch.float32))
    model = model.to("cuda").eval()
    surf, static, atmos, lat, lon, times = helpers.case_inputs(case, model.config)
    f = lambda d: {k: v.float() for k, v in d.items()}  # noqa: E731
    batch = Batch(f(surf), f(static), f(atmos), Metadata(lat.float(), lon.float(), times, tuple(case["levels"])))
    return model, batch


@pytest.mark.parametrize("name,H,W,world", [("base_pad", None, None, 2), ("base_pad", 192, 96, 2),
                                            ("base_pad", 192, 96, 3), ("base_pad", 192, 96, 4)])
@pytest.mark.parametrize("autocast", [False, True])
def test_sharded_equals_unsharded(name, H, W, world, autocast):
    model, batch = make(name, autocast, H, W)
    with torch.inference_mode():
        ref = model.forward(batch)
        bands, n_ex = run_virtual_ranks(model, batch, world)
    torch.cuda.synchronize()
    assert n_ex > 0
    assert [b.band for b in bands] == sorted(b.band for b in bands) and bands[0].band[0] == 0
    assert all(isinstance(b, BandBatch) and b.metadata.rollout_step == 1 for b in bands)
    surf, atmos = stitch(bands)
    for k, v in ref.surf_vars.items():
        assert surf[k].shape == v.shape
        assert helpers.rel_err(surf[k].cpu(), v.cpu()) < 2e-6, k
    for k, v in ref.atmos_vars.items():
        assert helpers.rel_err(atmos[k].cpu(), v.cpu()) < 2e-6, k
    assert torch.equal(torch.cat([b.metadata.lat for b in bands]), ref.metadata.lat)


@pytest.mark.parametrize("name", ["air_pollution", "wave", "stabilised_12h"])
def test_sharded_variants_equal_unsharded(name):
    """The variant hooks (difference prediction against the previous state, density / direction
    channels, water-body mask) address band-local planes correctly."""
    model, batch = make(name, False, 97, 96)
    with torch.inference_mode():
        ref = model.forward(batch)
        bands, _ = run_virtual_ranks(model, batch, 2)
    surf, atmos = stitch(bands)
    assert tuple(surf) == tuple(ref.surf_vars)
    for got, want in ((surf, ref.surf_vars), (atmos, ref.atmos_vars)):
        for k, v in want.items():
            a, b, flipped = helpers.nan_agreement(got[k].cpu(), v.cpu())
            assert flipped <= 1e-3 and helpers.rel_err(a, b) < 5e-6, (k, flipped)


def test_sharded_rollout_stays_distributed():
    """Two steps with the state kept as BandBatches (what rollout() does with gather_output=False)."""
    model, batch = make("base_pad", False, 192, 96)
    world = 2
    with torch.inference_mode():
        ref = list(aurora_amd.rollout(model, batch, steps=2))[1]
        bands1, _ = run_virtual_ranks(model, batch, world)
        dev = batch.crop(model.patch_size).to("cuda")
        nxt = []
        for r, b in enumerate(bands1):
            model._shard = Shard(r, world, None, gather_output=False)
            mine = Engine(model).local_band(dev)
            model._shard = None
            nxt.append(dataclasses.replace(
                b, surf_vars={k: torch.cat([mine.surf_vars[k][:, 1:], v], dim=1) for k, v in b.surf_vars.items()},
                atmos_vars={k: torch.cat([mine.atmos_vars[k][:, 1:], v], dim=1) for k, v in b.atmos_vars.items()}))
            assert isinstance(nxt[-1], BandBatch)
        bands2, _ = run_virtual_ranks(model, nxt, world)
    surf, atmos = stitch(bands2)
    assert all(b.metadata.rollout_step == 2 for b in bands2)
    for k, v in ref.atmos_vars.items():
        assert helpers.rel_err(atmos[k].cpu(), v.cpu()) < 2e-6, k


def test_gather_rows_and_band_output_limit():
    from aurora_amd.engine import lib

    src = torch.randn(50, 64, device="cuda")
    idx = torch.tensor([3, 3, 49, 0, 17], dtype=torch.int32, device="cuda")
    out = lib.gather_rows(src, idx, torch.empty(5, 64, device="cuda"))
    torch.cuda.synchronize()
    assert torch.equal(out, src[idx.long()])


def _proc(rank, world, port, ret):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model, batch = make("base_pad", False, 192, 96)
    with torch.inference_mode():
        ref = model.forward(batch) if rank == 0 else None
        model.configure_sharding(rank, world, gather_output=True)
        full = model.forward(batch)                       # every rank gets the whole prediction
        model.configure_sharding(rank, world, gather_output=False)
        preds = list(aurora_amd.rollout(model, batch, steps=2))  # state stays distributed
    torch.cuda.synchronize()
    assert isinstance(preds[1], BandBatch) and preds[1].metadata.rollout_step == 2
    if rank == 0:
        err = max(helpers.rel_err(full.atmos_vars[k].cpu(), v.cpu()) for k, v in ref.atmos_vars.items())
        err = max(err, max(helpers.rel_err(full.surf_vars[k].cpu(), v.cpu()) for k, v in ref.surf_vars.items()))
        h0, h1 = preds[0].band
        P = model.patch_size
        err_band = max(helpers.rel_err(preds[0].atmos_vars[k].cpu(), v[..., h0 * P:h1 * P, :].cpu())
                       for k, v in ref.atmos_vars.items())
        ret.put((err, err_band, tuple(full.surf_vars["2t"].shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_multiprocess_sharding_on_one_gpu(world):
    """The production code path (configure_sharding + forward / rollout, torch.distributed P2P halo
    exchange, broadcast gather) with real processes; transport = gloo with host staging because
    several ranks share this box's single GPU (RCCL needs one GPU per rank)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_proc, args=(r, world, 29700 + world, ret)) for r in range(world)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(timeout=300)
        assert p_.exitcode == 0
    err, err_band, shape = ret.get(timeout=10)
    assert shape == (1, 1, 192, 96)
    assert err < 2e-6 and err_band < 2e-6
