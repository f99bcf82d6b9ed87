"""Per-launch times of the un-sharded 0.25-degree step, grouped by (kernel kind, algorithmic work) = by shape.
    python tools/step_shapes.py [ModelClass HxW]          (GPU box)"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

cls = sys.argv[1] if len(sys.argv) > 1 else "AuroraPretrained"
H, W = (int(x) for x in sys.argv[2].split("x")) if len(sys.argv) > 2 else (721, 1440)
model = bench.build_model("cuda", cls)
batch = bench.synthetic_batch(model.config, H, W, 1, "cuda").crop(model.patch_size)
with torch.inference_mode():
    for _ in range(3):
        model.forward(batch)
    eng = model.engine()
    eng.profile_start()
    model.forward(batch)
    launches = eng.native.profile_end_list()
# A guarded linear is issued as TWO launches (two-term + three-term kernels) of which the device skips one: a launch of
# < 8 us with the work of a GEMM that takes far longer is such a skipped half -- listed apart, not averaged into its twin.
shapes = {}
for kind, ms, work in launches:
    skipped = kind.startswith("linear") and ms < 0.008 and work > 1e10
    key = (kind + (" [skipped half of a guarded pair]" if skipped else ""), work)
    n, t = shapes.get(key, (0, 0.0))
    shapes[key] = (n + 1, t + ms)
total = sum(ms for _, ms, _ in launches)
print(f"# {cls} {H}x{W}: {len(launches)} launches, {total:.2f} ms of kernels")
for (k, w), (n, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
    if t < 0.05 and "skipped" not in k:
        continue
    extra = f"{w * n / t / 1e9:8.0f} TFLOP/s" if k.startswith("linear") and t > 0 else (f"{w * n / t / 1e6:8.0f} GB/s" if w else "")
    print(f"{k:58s} work {w:16.0f}  x{n:3d}  mean {t / n * 1e3:9.1f} us  total {t:7.3f} ms  {extra}")
