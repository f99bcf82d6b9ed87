"""Per-kernel-kind time of ONE rank of an R-way latitude-band split (run alone: the handle's per-launch HIP events are then
meaningful), next to 1/R-th... of the un-sharded step's.      python tools/rank_kinds.py [R] [rank]     (GPU box)"""
import json
import sys
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


R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else R // 2
model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda").crop(model.patch_size)
with torch.inference_mode():
    model.forward(batch)
    e1 = model.engine()
    e1.profile_start()
    model.forward(batch)
    single = e1.profile_stop()
    model._shard = Shard(rank, R, None, gather_output=False)
    eng = Engine(model, transport=NoTransport(None, "cuda"))
    model._shard = None
    band = eng.local_band(batch)
    eng.step(band)
    eng.profile_start()
    eng.step(band)
    mine = eng.profile_stop()
    eng.profile_start()
    eng.step(band)
    launches = eng.native.profile_end_list()
share = (band.band[1] - band.band[0]) / band.full_patch_rows
rows = []
for k in sorted(set(single) | set(mine), key=lambda k_: -mine.get(k_, {"ms": 0})["ms"]):
    s, t = single.get(k, {"ms": 0.0, "launches": 0}), mine.get(k, {"ms": 0.0, "launches": 0})
    rows.append({"kind": k, "rank_ms": round(t["ms"], 3), "launches": t["launches"], "share_of_unsharded_ms": round(s["ms"] * share, 3),
                 "ratio": round(t["ms"] / (s["ms"] * share), 2) if s["ms"] else None})
# per shape (kind + algorithmic work identify it): launches, mean microseconds, TFLOP/s for the linears
shapes = {}
for kind, ms, work in launches:
    n, t = shapes.get((kind, work), (0, 0.0))
    shapes[(kind, work)] = (n + 1, t + ms)
by_shape = [{"kind": k, "work": w, "launches": n, "mean_us": round(t / n * 1e3, 1), "total_ms": round(t, 3),
             **({"tflops": round(w * n / t / 1e9, 0)} if k.startswith("linear") and t > 0 else {})}
            for (k, w), (n, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1])]
print(json.dumps({"by_shape": by_shape[:40]}), file=sys.stderr)
print(json.dumps({"ranks": R, "rank": rank, "row_share": share, "rank_total_ms": round(sum(v["ms"] for v in mine.values()), 2),
                  "share_of_unsharded_total_ms": round(sum(v["ms"] for v in single.values()) * share, 2), "per_kind": rows}, indent=1))
