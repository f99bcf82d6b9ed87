"""Per-shape launch times inside one forecast step (0.25 degree): the step is sequenced from Python
(AURORA_NATIVE_STEP=0, same kernels as the C-ABI handle) with a HIP event pair around every launch, and launches
are grouped by (kernel, algorithmic work) -- for the GEMMs that is one group per (M, N, K)."""
import os
import sys
from collections import defaultdict
from pathlib import Path

os.environ["AURORA_NATIVE_STEP"] = "0"
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from aurora_amd.engine import lib  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
batch = bench.synthetic_batch(model.config, 721, 1440, 1, dev)
with torch.inference_mode():
    for _ in range(2):
        model.forward(batch)
    torch.cuda.synchronize()
    lib.profile_start(None)
    model.forward(batch)
    rec, lib._profile = lib._profile, None
    torch.cuda.synchronize()
groups = defaultdict(list)
for name, work, e0, e1 in rec:
    groups[(name, work)].append(e0.elapsed_time(e1))
rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
total = sum(sum(v) for v in groups.values())
print(f"{'kernel':28s} {'work':>12s} {'n':>4s} {'mean ms':>9s} {'total ms':>9s} {'rate':>10s}")
for (name, work), v in rows[:40]:
    t = sum(v) / len(v)
    rate = f"{work / t / 1e9:8.1f} T" if name.startswith("linear") and work else (f"{work / t / 1e6:8.1f} G" if work else "")
    print(f"{name:28s} {work:12.4g} {len(v):4d} {t:9.3f} {sum(v):9.2f} {rate:>10s}")
print(f"total {total:.1f} ms")
