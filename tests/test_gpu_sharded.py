"""Latitude-band sharding on ONE GPU: R handles of the same model run as virtual ranks in one process, one thread
each; their halo exchanges are carried out by copying exactly the staging buffers the handle's `post` callback names
(the RCCL point-to-point path moves the same bytes between processes).  The stitched bands must equal the un-sharded
engine."""
import dataclasses
import queue
import subprocess
from pathlib import Path
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
    model_ref, _ = make("base_pad", False, 192, 96)    # an un-sharded twin (same seeded weights) for the roll-out reference
    with torch.inference_mode():
        ref = model.forward(batch) if rank == 0 else None
        model.configure_sharding(rank, world, gather_output=True)
        full = model.forward(batch)                       # every rank gets the whole prediction
        # a gathered prediction carries the caller's static fields (the next roll-out step, `_to_host` and `write_rollout`
        # read them from it), and a gathered roll-out runs more than one step
        assert set(full.static_vars) == set(batch.static_vars)
        assert all(torch.equal(full.static_vars[k].cpu(), v.cpu()) for k, v in batch.static_vars.items())
        ref2 = list(aurora_amd.rollout(model_ref, batch, steps=2))[1] if rank == 0 else None
        gathered2 = list(aurora_amd.rollout(model, batch, steps=2))[1]
        assert not isinstance(gathered2, BandBatch) and set(gathered2.static_vars) == set(batch.static_vars)
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
        err = max(err, max(helpers.rel_err(gathered2.atmos_vars[k].cpu(), v.cpu()) for k, v in ref2.atmos_vars.items()))
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


_RCCL_LOOP = r'''
import os, sys
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29761", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.all_reduce(torch.ones(1, device="cuda"))      # the communicator is created here
    torch.cuda.synchronize()
except Exception as e:                                  # no RCCL on this box: not what this test is about
    print("RCCL-UNAVAILABLE", repr(e)); sys.exit(77)
from aurora_amd.engine import lib, native

class OneRank:                                          # what _Transport reads of a Shard
    rank, world, group = 0, 1, None

tr = native._Transport(OneRank(), "cuda")
n = 1 << 20
tr.allocate(2 * n)
pattern = (torch.arange(n, device="cuda") % 251).to(torch.uint8)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):                                 # ~10^13 flop queued AHEAD of the fill: the host runs far in front
        a = (a @ a) * 1e-2
    tr.send[:n] = pattern + (a[0, 0] * 0).to(torch.uint8)     # ... and the fill depends on that work
    tr.recv[:n] = 0xEE
    msgs = (lib.HipHaloMsg * 1)()
    msgs[0].peer, msgs[0].reserved, msgs[0].offset, msgs[0].bytes = 0, 0, 0, n
    assert tr._post(None, msgs, 1, msgs, 1, s.cuda_stream) == 0, tr.error   # send to / receive from rank 0 = this rank
    assert tr._wait(None, s.cuda_stream) == 0, tr.error
    got = tr.recv[:n].clone()                           # ordered behind the receive by `wait`, on the launch stream
    tr.send[:n] = 0                                     # overwriting the staging buffer AFTER the send was ordered
    # the stream assertion: a post on a stream that is not torch's current one must be refused
    bad = tr._post(None, msgs, 1, msgs, 1, torch.cuda.default_stream().cuda_stream)
s.synchronize()
assert bad == -1 and "current stream" in str(tr.error), (bad, tr.error)
assert torch.equal(got, pattern), int((got != pattern).sum())
print("RCCL-LOOP-OK", tr.exchanges)
dist.destroy_process_group()
'''


def test_rccl_carries_a_halo_message_in_stream_order(tmp_path):
    """The RCCL branch of the halo transport (`_Transport._post`: `batch_isend_irecv` on the NCCL backend, `wait` on the
    launch stream) on the one GPU there is: a one-rank communicator, the rank its own neighbour -- RCCL's send / receive
    kernels move the message, ordered behind a producer that is still running when the host posts and in front of the
    consumer.  What it cannot show is a second GPU; what it does show is that the bytes, the grouping and the stream
    ordering of the production transport are right.  In its own process, so that a hung collective cannot hang the suite."""
    import os
    import subprocess
    import sys
    script = tmp_path / "rccl_loop.py"
    script.write_text(_RCCL_LOOP)
    root = str(Path(__file__).resolve().parents[1])
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    if r.returncode == 77:
        pytest.skip("RCCL cannot initialise on this box: " + r.stdout.strip()[-300:])
    assert r.returncode == 0 and "RCCL-LOOP-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_c_rccl_transport_loops_a_message_back(tmp_path):
    """The plain-C halo transport (examples/c_host/rccl_transport.c: ncclSend / ncclRecv in one ncclGroup on a side stream,
    two events against the launch stream) as a one-rank loop-back, examples/c_host/rccl_loopback.c: no Python, no torch in
    the process.  Together with tests/test_c_host.py (the band mode of the C host links against it) this is what one GPU
    can verify of the C transport."""
    root = Path(__file__).resolve().parents[1]
    src = root / "examples" / "c_host"
    exe = tmp_path / "rccl_loopback"
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-D_POSIX_C_SOURCE=200809L", "-D__HIP_PLATFORM_AMD__", f"-I{root / 'include'}",
           f"-I{src}", "-I/opt/rocm/include", str(src / "rccl_loopback.c"), str(src / "rccl_transport.c"), "-L/opt/rocm/lib",
           "-lamdhip64", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    r = subprocess.run([str(exe), str(tmp_path / "nccl_id")], capture_output=True, text=True, timeout=240)
    if r.returncode == 77:
        pytest.skip("RCCL cannot initialise on this box: " + r.stderr.strip()[-300:])
    assert r.returncode == 0 and "RCCL-LOOPBACK-OK exchanges=1" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
