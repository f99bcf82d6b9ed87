"""List the fp32 linears of one 0.25-degree step with their shapes and times (HIP events)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from aurora_amd.engine import lib  # noqa: E402

model = bench.build_model("cuda")
batch = bench.synthetic_batch(model.config, 721, 1440, 1, "cuda")
rec = []
orig = lib.linear


def wrapped(a, w, bias, out, **kw):
    if a.dtype != torch.float32:
        return orig(a, w, bias, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(a, w, bias, out, **kw)
    e1.record()
    n = kw.get("n") or w.shape[0]
    k = kw.get("k") or a.shape[1]
    rec.append((a.shape[0], n, k, kw.get("act", 0), kw.get("residual") is not None, kw.get("out2") is not None, e0, e1))
    return r


with torch.inference_mode():
    model.forward(batch)
    model.forward(batch)
    lib.linear = wrapped
    import aurora_amd.engine.engine as eng
    eng.lib.linear = wrapped
    model.forward(batch)
    torch.cuda.synchronize()
tot = 0.0
for M, N, K, act, res, o2, e0, e1 in rec:
    ms = e0.elapsed_time(e1)
    tot += ms
    if ms > 0.2:
        print(f"M={M:7d} N={N:5d} K={K:5d} act={act} res={int(res)} out2={int(o2)}  {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s")
print(f"total fp32 linear time {tot:.1f} ms in {len(rec)} calls")
