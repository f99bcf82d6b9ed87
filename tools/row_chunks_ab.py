"""Row chunks of the token-local half of a backbone block on concurrent HIP streams (csrc/step.hip) -- A/B inside the step.

For every setting of (AURORA_ROW_CHUNKS, AURORA_CHUNK_SYNC) a fresh handle steps the 0.25-degree configuration of
bench.py; reported: ms per step (un-instrumented), the per-kernel event sums of one more step (a sum ABOVE the step time
is the overlap), and whether the prediction equals the un-chunked handle's bit for bit (the chunks run the same launches
over the same rows: it must).      python tools/row_chunks_ab.py [--grid 721x1440] [--steps 10] [settings ...]   (GPU box)
a setting is "chunks,sync[,min_rows]", e.g. 2,2 3,0 2,1,1024"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def run(model, batch, steps):
    model._engine = None   # a new handle reads the environment
    with torch.inference_mode():
        for _ in range(3):
            pred = model.forward(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pred = model.forward(batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        eng = model.engine()
        eng.native.profile_begin()
        model.forward(batch)
        kinds = eng.native.profile_end()
    out = {**pred.surf_vars, **{"a." + k: v for k, v in pred.atmos_vars.items()}}
    return ms, {k: v["ms"] for k, v in kinds.items() if v["launches"]}, {k: v.clone() for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="721x1440")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--model", default="AuroraPretrained")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("settings", nargs="*", default=["1,0", "2,2", "2,1", "2,0", "3,2", "3,0"])
    args = ap.parse_args()
    H, W = map(int, args.grid.split("x"))
    model = bench.build_model("cuda", args.model)
    batch = bench.synthetic_batch(model.config, H, W, 1, "cuda").crop(model.patch_size)
    base = None
    for rep in range(args.reps):
        for s in args.settings:
            p = s.split(",")
            os.environ["AURORA_ROW_CHUNKS"] = p[0]
            os.environ["AURORA_CHUNK_SYNC"] = p[1]
            if len(p) > 2:
                os.environ["AURORA_CHUNK_MIN_ROWS"] = p[2]
            else:
                os.environ.pop("AURORA_CHUNK_MIN_ROWS", None)
            ms, kinds, out = run(model, batch, args.steps)
            if base is None:
                base = out
            same = all(torch.equal(out[k], base[k]) for k in base)
            worst = max(float((out[k] - base[k]).abs().max() / base[k].abs().max()) for k in base)
            print(json.dumps({"setting": s, "ms_per_step": round(ms, 2), "kernel_sum_ms": round(sum(kinds.values()), 2),
                              "bit_identical_to_first": same, "max_rel_diff": worst,
                              "kinds": {k: round(v, 2) for k, v in kinds.items()}}), flush=True)


if __name__ == "__main__":
    main()
