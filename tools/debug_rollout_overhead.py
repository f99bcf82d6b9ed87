"""Where does the eager roll-out lose time against back-to-back forwards?  (one GPU, 0.25 degree)"""
import dataclasses
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from aurora_amd import rollout  # noqa: E402

model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda")


def timeit(name, fn, n=5):
    with torch.inference_mode():
        fn(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = fn(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{name:46s} {1e3 * (t2 - t0) / n:7.1f} ms/step   host-side {1e3 * (t1 - t0) / n:7.1f} ms/step", flush=True)


def same_batch(n):
    for _ in range(n):
        model.forward(batch)


def cropped_same(n):
    b = batch.type(torch.float32).crop(4).to("cuda")
    for _ in range(n):
        model.forward(b)


def manual_cat(n):
    b = batch.type(torch.float32).crop(4).to("cuda")
    for _ in range(n):
        pred = model.forward(b)
        b = dataclasses.replace(
            pred,
            surf_vars={k: torch.cat([b.surf_vars[k][:, 1:], v], dim=1) for k, v in pred.surf_vars.items()},
            atmos_vars={k: torch.cat([b.atmos_vars[k][:, 1:], v], dim=1) for k, v in pred.atmos_vars.items()})


def roll(n):
    for _ in rollout(model, batch, steps=n):
        pass


timeit("forward, same (uncropped) batch", same_batch)
timeit("forward, same cropped batch", cropped_same)
timeit("forward + history cat (= rollout body)", manual_cat)
timeit("rollout()", roll)
