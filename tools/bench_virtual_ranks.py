"""Compute-side cost of latitude-band sharding, measured on ONE GPU: the R ranks of a sharded 0.25-degree step are
run as virtual ranks in one process (tests/test_gpu_sharded.py harness; halo rows copied device-to-device), so the
total GPU time is the SUM of the per-rank compute.  sum / R is what one rank of an R-GPU job computes per step
(communication excluded); (un-sharded step) / sum is the compute efficiency of the partition (halo recomputation,
thin GEMMs, tile quantisation).

    python tools/bench_virtual_ranks.py [R ...]      (default 2 4 8)
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from tests.test_gpu_sharded import make_engines, run_virtual_ranks  # noqa: E402

model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda").crop(model.patch_size)


def timed(fn, n=3):
    with torch.inference_mode():
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


single = timed(lambda: model.forward(batch))
print(json.dumps({"ranks": 1, "ms_per_step": single}), flush=True)
for R in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    engines = make_engines(model, R)
    total = timed(lambda: run_virtual_ranks(model, batch, R, engines), n=2)
    del engines
    print(json.dumps({"ranks": R, "sum_of_rank_compute_ms": total, "per_rank_ms": total / R,
                      "compute_efficiency": single / total}), flush=True)
