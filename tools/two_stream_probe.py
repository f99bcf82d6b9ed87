"""Do two half-forecasts on two HIP streams overlap on one GPU?  The two latitude bands of a 2-way split are stepped
(a) one after the other on one stream and (b) concurrently, one thread and one stream each (no halo transport: timing only).
If (b) beats the un-sharded step, interleaving two independent row-halves of a forecast would hide HBM-bound kernels
(LayerNorm, attention) under MFMA-bound ones (GEMMs).      python tools/two_stream_probe.py      (GPU box)"""
import json
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from aurora_amd.engine import native  # noqa: E402
from aurora_amd.engine.engine import Engine, Shard  # noqa: E402


class NoTransport(native._Transport):
    def allocate(self, n_bytes):
        super().allocate(n_bytes)
        self.send.zero_()
        self.recv.zero_()

    def _post(self, *a):
        return 0

    def _wait(self, *a):
        return 0


model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda").crop(model.patch_size)
N = 6
with torch.inference_mode():
    for _ in range(3):
        model.forward(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        model.forward(batch)
    torch.cuda.synchronize()
    single = (time.perf_counter() - t0) / N * 1e3
engs, bands = [], []
for r in range(2):
    model._shard = Shard(r, 2, None, gather_output=False)
    engs.append(Engine(model, transport=NoTransport(None, "cuda")))
    model._shard = None
    bands.append(engs[-1].local_band(batch))
with torch.inference_mode():
    for e, b in zip(engs, bands):
        e.step(b)
        e.step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        for e, b in zip(engs, bands):
            e.step(b)
    torch.cuda.synchronize()
    serial = (time.perf_counter() - t0) / N * 1e3
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(r, delay):
    with torch.inference_mode(), torch.cuda.stream(streams[r]):
        if delay:
            time.sleep(delay)
        for _ in range(N):
            engs[r].step(bands[r])


res = {"unsharded_ms": single, "two_bands_serial_ms": serial}
for delay in (0.0, 0.02):
    torch.cuda.synchronize()
    th = [threading.Thread(target=run, args=(r, delay * r)) for r in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    res[f"two_bands_two_streams_ms(delay {delay})"] = ((time.perf_counter() - t0) - delay) / N * 1e3
print(json.dumps(res))
