"""Where the compute of a latitude-band split goes, per kernel kind: every launch of the un-sharded 0.25-degree step and of
the R virtual ranks of the same step is bracketed by HIP events inside the handle (aurora_hip_profile_begin / _end) and
summed per kind over the ranks.  (sum over ranks) / (un-sharded) per kind is that kind's share of the efficiency loss.

    python tools/virtual_rank_kinds.py [R]      (default 8; run on the GPU box)
"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from tests.test_gpu_sharded import make_engines, run_virtual_ranks  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda").crop(model.patch_size)
with torch.inference_mode():
    model.forward(batch)
    eng = model.engine()
    eng.profile_start()
    model.forward(batch)
    single = eng.profile_stop()
    engines = make_engines(model, R)
    run_virtual_ranks(model, batch, R, engines)          # warm-up: plans, staging, workspace
    for e in engines:
        e.profile_start()
    run_virtual_ranks(model, batch, R, engines)
    ranks = [e.profile_stop() for e in engines]
total = {}
for prof in ranks:
    for k, v in prof.items():
        d = total.setdefault(k, {"launches": 0, "ms": 0.0})
        d["launches"] += v["launches"]
        d["ms"] += v["ms"]
rows = []
for k in sorted(set(single) | set(total), key=lambda k_: -(total.get(k_, {}).get("ms", 0.0))):
    s, t = single.get(k, {"launches": 0, "ms": 0.0}), total.get(k, {"launches": 0, "ms": 0.0})
    rows.append({"kind": k, "single_ms": round(s["ms"], 2), "single_launches": s["launches"], f"sum_{R}_ranks_ms": round(t["ms"], 2),
                 "launches": t["launches"], "ratio": round(t["ms"] / s["ms"], 2) if s["ms"] else None})
print(json.dumps({"ranks": R, "single_total_ms": round(sum(v["ms"] for v in single.values()), 2),
                  "sum_total_ms": round(sum(v["ms"] for v in total.values()), 2), "per_kind": rows}, indent=1))
