"""Secondary measurements on ONE MI355X for the other BASELINE.json configurations (bench.py owns configs[1]):

    python tools/bench_configs.py [graph] [highres] [airpollution]

  graph         configs[2]: 0.25-degree roll-out with the step replayed from a hipGraph (ms per step over 8 steps)
  highres       configs[3] on one GPU: AuroraHighRes (patch 10, LoRA), 1801 x 3600 -- does it fit, how fast
  airpollution  configs[4] on one GPU: AuroraAirPollution, 451 x 900, 12 h steps, eager roll-out

Weights are random (seed 0), inputs synthetic (seed 1), exactly as in bench.py.  Prints one JSON line per case.
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

import aurora_amd  # noqa: E402
from aurora_amd import rollout  # noqa: E402


def build(cls, **kw):
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = cls(autocast=True, **kw)
        with torch.no_grad():
            for p in model.parameters():
                if not p.any():
                    p.normal_(std=0.02)
    return model.eval()


def timed_rollout(model, batch, steps, **kw):
    with torch.inference_mode():
        gen = rollout(model, batch, steps=steps + 2, **kw)
        next(gen), next(gen)                     # warm-up (graph capture included)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for pred in gen:
            pass
        torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in pred.surf_vars.values() if v.numel())
    return (time.perf_counter() - t0) / steps * 1e3


def kinds(model, batch):
    """ms per kernel family of ONE step (HIP events around every launch inside the handle)."""
    eng = model.engine()
    with torch.inference_mode():
        eng.profile_start()
        model.forward(batch)
        prof = eng.profile_stop()
    return {k: round(v["ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}


def positive_batch(cfg, H, W, levels):
    """Like bench.synthetic_batch; variables the model treats as positive get |randn| (aurora.py:733-742 upstream)."""
    b = bench.synthetic_batch(cfg, H, W, 1, "cuda", levels=levels)
    pos = set(cfg.positive_surf_vars) | set(cfg.positive_atmos_vars)
    from aurora_amd import normalisation as nz
    for d in (b.surf_vars, b.atmos_vars):
        for k in d:
            if k in pos:
                d[k] = d[k].abs()
    return b


cases = sys.argv[1:] or ["graph", "highres", "airpollution"]
def mem(model=None):
    """GiB: peak of torch's allocator (inputs, history, weights), the handle's own step workspace (hipMalloc'ed, not seen
    by torch) and what the device reports as in use right now (everything, incl. the handle's weight copies)."""
    free, total = torch.cuda.mem_get_info()
    out = {"torch_peak_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "device_in_use_GiB": round((total - free) / 2 ** 30, 1)}
    if model is not None:
        out["handle_workspace_GiB"] = round(model.engine().native.workspace_bytes() / 2 ** 30, 1)
    return out
if "graph" in cases:
    model = build(aurora_amd.AuroraPretrained)
    batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda")
    n = 40   # BASELINE configs[2]: 40-step roll-out
    eager = timed_rollout(model, batch, n)
    graphed = timed_rollout(model, batch, n, graph=True)
    eager_host = timed_rollout(model, batch, n, to_host=True)       # every prediction delivered in pinned host memory
    graphed_host = timed_rollout(model, batch, n, graph=True, to_host=True)
    print(json.dumps({"case": f"configs[2] {n}-step rollout, 0.25deg, 1 GPU", "ms_per_step_eager": eager,
                      "ms_per_step_hipgraph": graphed, "ms_per_step_eager_to_host": eager_host,
                      "ms_per_step_hipgraph_to_host": graphed_host,
                      "to_host_overhead": graphed_host / graphed - 1.0, **mem(model)}), flush=True)
    del model, batch
    torch.cuda.empty_cache()
if "highres" in cases:
    torch.cuda.reset_peak_memory_stats()
    model = build(aurora_amd.AuroraHighRes)
    batch = bench.synthetic_batch(model.config, 1801, 3600, 1, "cuda")
    ms = timed_rollout(model, batch, 3)
    print(json.dumps({"case": "configs[3] AuroraHighRes 0.1deg 1801x3600 on ONE GPU (LoRA step >= 1)", "ms_per_step": ms,
                      **mem(model), "kernel_ms_per_step": kinds(model, batch)}), flush=True)
    del model, batch
    torch.cuda.empty_cache()
if "airpollution" in cases:
    torch.cuda.reset_peak_memory_stats()
    model = build(aurora_amd.AuroraAirPollution)
    batch = positive_batch(model.config, 451, 900, bench.LEVELS)
    ms = timed_rollout(model, batch, 4)
    print(json.dumps({"case": "configs[4] AuroraAirPollution 0.4deg 451x900, 12 h steps, 1 GPU", "ms_per_step": ms,
                      **mem(model), "kernel_ms_per_step": kinds(model, batch)}), flush=True)
