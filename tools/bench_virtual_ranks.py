"""Compute-side cost of latitude-band sharding, measured on ONE GPU: each of the R ranks of a sharded 0.25-degree step is
run ALONE (its handle, its band of the inputs; the halo transport is a no-op that leaves zeros in the receive buffer, so
the rank does exactly the work it would do -- gather, qkv GEMM, halo projection, interior / boundary windows -- on halo
values that do not matter for timing).  sum over ranks = the GPU time of the whole sharded step without communication;
(un-sharded step) / sum is the compute efficiency of the partition (halo work, thin GEMMs, tile quantisation);
max over ranks is what bounds a real R-GPU step from below.

    python tools/bench_virtual_ranks.py [--model CLS --grid HxW] [R ...]      (default: AuroraPretrained 721x1440, R = 2 4 8)

(tests/test_gpu_sharded.py runs the ranks TOGETHER, one thread each, with real halo copies: that checks results; its wall
time includes the threads' waiting for each other and is not a compute measurement.)
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
    def allocate(self, n_bytes):
        super().allocate(n_bytes)
        self.send.zero_()
        self.recv.zero_()

    def _post(self, *a):
        return 0

    def _wait(self, *a):
        return 0


def timed(fn, n=4):
    with torch.inference_mode():
        fn()
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


argv, cls_name, grid = sys.argv[1:], "AuroraPretrained", (721, 1440)
while argv and argv[0].startswith("--"):
    if argv[0] == "--model":
        cls_name = argv[1]
    elif argv[0] == "--grid":
        grid = tuple(int(x) for x in argv[1].split("x"))
    else:
        raise SystemExit(f"unknown option {argv[0]}")
    argv = argv[2:]
model = bench.build_model("cuda", cls_name)
batch = bench.synthetic_batch(model.config, grid[0], grid[1], 1, "cuda").crop(model.patch_size)
single = timed(lambda: model.forward(batch))
print(json.dumps({"model": cls_name, "grid": list(grid), "ranks": 1, "ms_per_step": single}), flush=True)
for R in [int(a) for a in argv] or [2, 4, 8]:
    per_rank = []
    for r in range(R):
        model._shard = Shard(r, R, None, gather_output=False)
        eng = Engine(model, transport=NoTransport(None, "cuda"))
        model._shard = None
        band = eng.local_band(batch)
        per_rank.append(timed(lambda: eng.step(band)))
        del eng, band
        torch.cuda.empty_cache()
    total = sum(per_rank)
    print(json.dumps({"ranks": R, "per_rank_ms": [round(x, 2) for x in per_rank], "sum_of_rank_compute_ms": total,
                      "max_rank_ms": max(per_rank), "compute_efficiency": single / total,
                      "strong_scaling_bound_without_communication": single / (R * max(per_rank))}), flush=True)
