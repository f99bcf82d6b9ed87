"""Host time of a step: how long `Aurora.forward` (= ctypes -> aurora_hip_step: a dry walk of the step for the workspace,
then ~750 launches from C++) occupies the calling thread, against the device time of the step -- un-sharded, and for one
rank of R latitude bands (one rank at a time here, its neighbours' halos not exchanged: `post` / `wait` are no-ops, so the
numbers are the handle's own host cost without a transport).

    python tools/host_time.py [R ...]       (default 8; run on the GPU box)
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from aurora_amd.engine import native  # noqa: E402
from aurora_amd.engine.engine import Engine, Shard  # noqa: E402


class NoTransport(native._Transport):
    def _post(self, *a):
        return 0

    def _wait(self, *a):
        return 0


def measure(step, n=10):
    with torch.inference_mode():
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        host = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            h0 = time.perf_counter()
            step()
            host += time.perf_counter() - h0
            torch.cuda.synchronize()      # (so that the host never waits for a full launch queue)
        total = time.perf_counter() - t0
    return {"host_ms_per_step": host / n * 1e3, "step_ms": total / n * 1e3}


model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda").crop(model.patch_size)
print(json.dumps({"ranks": 1, **measure(lambda: model.forward(batch))}), flush=True)
for R in [int(a) for a in sys.argv[1:]] or [8]:
    model._shard = Shard(R // 2, R, None, gather_output=False)
    eng = Engine(model, transport=NoTransport(None, "cuda"))
    model._shard = None
    band = eng.local_band(batch)
    print(json.dumps({"ranks": R, "rank": R // 2, **measure(lambda: eng.step(band))}), flush=True)
