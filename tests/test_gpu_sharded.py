"""Latitude-band sharding on ONE GPU: R handles of the same model run as virtual ranks in one process, one thread
each; their halo exchanges are carried out by copying exactly the staging buffers the handle's `post` callback names
(the RCCL point-to-point path moves the same bytes between processes).  The stitched bands must equal the un-sharded
engine."""
import dataclasses
import queue
import threading

import pytest
import torch

import aurora_amd
from aurora_amd import Batch, Metadata
from aurora_amd.batch import BandBatch
from aurora_amd.engine import native
from aurora_amd.engine.engine import Engine, Shard
from tests import helpers
from tests.golden_cases import CASES

pytestmark = pytest.mark.gpu


class VirtualTransport(native._Transport):
    """The handle's halo callbacks for ranks that live in one process: a message is a device-to-device copy from the
    sender's staging buffer into the receiver's.  All ranks launch on the same (default) stream, so "the copy is enqueued
    after the sender's gather kernel" is all the ordering needed: the sender posts AFTER launching its gather, the
    receiver copies in `wait`, and the sender's `wait` holds until its messages have been copied out (the next
    exchange's gather overwrites the staging buffer)."""

    TIMEOUT = 120.0

    def __init__(self, rank, hub) -> None:
        super().__init__(None, "cuda")
        self.rank, self.hub = rank, hub
        self.sent, self.expect = [], []

    def _post(self, user, sends, n_sends, recvs, n_recvs, stream) -> int:
        try:
            for i in range(n_sends):
                m = sends[i]
                assert abs(m.peer - self.rank) == 1
                consumed = threading.Event()
                self.hub.box(self.rank, m.peer).put((self.send[m.offset:m.offset + m.bytes], consumed))
                self.sent.append(consumed)
            self.expect = [(recvs[i].peer, recvs[i].offset, recvs[i].bytes) for i in range(n_recvs)]
            self.hub.exchanges += 1
            return 0
        except BaseException as e:  # noqa: BLE001
            self.error = e
            return -1

    def _wait(self, user, stream) -> int:
        try:
            for peer, offset, n_bytes in self.expect:
                src, consumed = self.hub.box(peer, self.rank).get(timeout=self.TIMEOUT)
                assert src.numel() == n_bytes, (src.numel(), n_bytes)
                self.recv[offset:offset + n_bytes].copy_(src)
                consumed.set()
            for consumed in self.sent:
                assert consumed.wait(self.TIMEOUT), "a halo message was never received"
            self.sent, self.expect = [], []
            return 0
        except BaseException as e:  # noqa: BLE001
            self.error = e
            return -1


class Hub:
    def __init__(self) -> None:
        self.boxes, self.lock, self.exchanges = {}, threading.Lock(), 0

    def box(self, src, dst):
        with self.lock:
            return self.boxes.setdefault((src, dst), queue.Queue())


def make_engines(model, world):
    hub = Hub()
    engines = []
    for r in range(world):
        model._shard = Shard(r, world, None, gather_output=False)
        engines.append(Engine(model, transport=VirtualTransport(r, hub)))
    model._shard = None
    return engines


def run_virtual_ranks(model, batch, world, engines=None):
    """One sharded step on `world` virtual ranks (threads); returns the per-rank BandBatches and the number of exchanges
    posted.  `batch`: the full batch, or a list with every rank's BandBatch."""
    engines = engines or make_engines(model, world)
    hub = engines[0].native.transport.hub
    before = hub.exchanges
    out, errors = [None] * world, []

    def run(r):
        try:
            with torch.inference_mode():
                out[r] = engines[r].step(batch if not isinstance(batch, list) else batch[r])
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    assert all(b.empty() for b in hub.boxes.values())
    return out, hub.exchanges - before


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
                                            ("base_pad", 192, 96, 3), ("base_pad", 192, 96, 4),
                                            # thin bands: searched partitions (csrc/band.hip:band_rows)
                                            ("base_pad", 256, 96, 4), ("base_pad", 256, 96, 5)])
@pytest.mark.parametrize("autocast", [False, True])
def test_sharded_equals_unsharded(name, H, W, world, autocast):
    """2e-6 is the bound to keep, not bit equality: fp32 steps are bit-identical band by band, but a bf16 band may take the
    split-K form of a few-tile / long-K linear (aurora_hip_linear_ws) where the un-sharded step does not, and K-slices added
    in fp32 before the one rounding to bf16 differ from the un-split product by up to one bf16 ulp of that linear's output
    (tests/test_gpu_ops.py::test_linear_ws_split_k_equals_the_unsplit_product; the production-width geometry of
    tests/test_gpu_production.py runs split-K end to end against the oracle)."""
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
        engines = make_engines(model, world)
        for r, b in enumerate(bands1):
            mine = engines[r].local_band(dev)
            nxt.append(dataclasses.replace(
                b, surf_vars={k: torch.cat([mine.surf_vars[k][:, 1:], v], dim=1) for k, v in b.surf_vars.items()},
                atmos_vars={k: torch.cat([mine.atmos_vars[k][:, 1:], v], dim=1) for k, v in b.atmos_vars.items()}))
            assert isinstance(nxt[-1], BandBatch)
        bands2, _ = run_virtual_ranks(model, nxt, world, engines)
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


def _proc(rank, world, port, ret, tmp):
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
        # sharded output: every rank writes ITS band of every step, nothing is gathered
        paths = aurora_amd.write_rollout(model, batch, 2, os.path.join(tmp, "pred.{step}.{rank:02d}.nc"))
    torch.cuda.synchronize()
    assert isinstance(preds[1], BandBatch) and preds[1].metadata.rollout_step == 2
    assert [os.path.basename(p_) for p_ in paths] == [f"pred.{s_}.{rank:02d}.nc" for s_ in (1, 2)]
    dist.barrier()
    if rank == 0:   # the per-rank files of step 1 reassemble to the un-sharded prediction
        joined = Batch.from_netcdf(os.path.join(tmp, "pred.1.{rank:02d}.nc"))
        assert joined.spatial_shape == ref.spatial_shape
        for k, v in ref.atmos_vars.items():
            assert helpers.rel_err(joined.atmos_vars[k], v.cpu()) < 2e-6, k
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
def test_multiprocess_sharding_on_one_gpu(world, tmp_path):
    """The production code path (configure_sharding + forward / rollout, torch.distributed P2P halo
    exchange, one all-gather of the packed prediction) with real processes; transport = gloo with host staging because
    several ranks share this box's single GPU (RCCL needs one GPU per rank)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_proc, args=(r, world, 29700 + world, ret, str(tmp_path))) for r in range(world)]
    for p_ in procs:
        p_.start()
    for p_ in procs:
        p_.join(timeout=300)
        assert p_.exitcode == 0
    err, err_band, shape = ret.get(timeout=10)
    assert shape == (1, 1, 192, 96)
    assert err < 2e-6 and err_band < 2e-6
