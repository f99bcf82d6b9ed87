"""Where the compute of an R-way latitude-band split goes: one step of R virtual ranks on one GPU (see
tools/bench_virtual_ranks.py) with a HIP event pair around every launch, summed per kernel and -- for the GEMMs -- per
(kernel, work) group, next to the un-sharded step sequenced the same way.

    python tools/virtual_rank_shapes.py [R]      (default 8)
"""
import os
import sys
from collections import defaultdict
from pathlib import Path

os.environ["AURORA_NATIVE_STEP"] = "0"
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from aurora_amd.engine import lib  # noqa: E402
from tests.test_gpu_sharded import make_engines, run_virtual_ranks  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda").crop(model.patch_size)


def profile(fn):
    with torch.inference_mode():
        fn()
        fn()
        torch.cuda.synchronize()
        lib.profile_start(None)
        fn()
        rec, lib._profile = lib._profile, None
        torch.cuda.synchronize()
    kinds, groups = defaultdict(float), defaultdict(list)
    for name, work, e0, e1 in rec:
        t = e0.elapsed_time(e1)
        kinds[name] += t
        groups[(name, work)].append(t)
    return kinds, groups


k1, g1 = profile(lambda: model.forward(batch))
engines = make_engines(model, R)
kR, gR = profile(lambda: run_virtual_ranks(model, batch, R, engines))
print(f"{'kernel':28s} {'1 rank ms':>10s} {'sum of ' + str(R) + ' ms':>12s} {'ratio':>6s}")
for name in sorted(kR, key=lambda n: -kR[n]):
    print(f"{name:28s} {k1.get(name, 0.0):10.2f} {kR[name]:12.2f} {kR[name] / k1[name] if k1.get(name) else 0:6.2f}")
print(f"{'total':28s} {sum(k1.values()):10.2f} {sum(kR.values()):12.2f}")
print("\nGEMM groups of the split (work = 2MNK of one launch):")
for (name, work), v in sorted(gR.items(), key=lambda kv: -sum(kv[1]))[:30]:
    if name.startswith("linear"):
        t = sum(v) / len(v)
        print(f"{name:12s} work {work:10.4g}  n {len(v):4d}  mean {t * 1e3:8.1f} us  total {sum(v):7.2f} ms  {work / t / 1e9:7.1f} TFLOP/s")
