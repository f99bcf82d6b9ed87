"""Find the fp32 linear that produces non-finite values in the README example (2 x fp16 split debugging)."""
import sys
from datetime import datetime
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import aurora_amd  # noqa: E402
from aurora_amd import Batch, Metadata  # noqa: E402
from aurora_amd.engine import lib  # noqa: E402
import aurora_amd.engine.engine as eng  # noqa: E402

model = aurora_amd.AuroraSmallPretrained()
torch.manual_seed(0)
for p in model.parameters():
    if p.abs().sum() == 0:
        torch.nn.init.normal_(p, std=0.02)
model = model.to("cuda").eval()
batch = Batch(
    surf_vars={k: torch.randn(1, 2, 17, 32) for k in ("2t", "10u", "10v", "msl")},
    static_vars={k: torch.randn(17, 32) for k in ("lsm", "z", "slt")},
    atmos_vars={k: torch.randn(1, 2, 4, 17, 32) for k in ("z", "u", "v", "t", "q")},
    metadata=Metadata(lat=torch.linspace(90, -90, 17), lon=torch.linspace(0, 360, 33)[:-1],
                      time=(datetime(2020, 6, 1, 12, 0),), atmos_levels=(100, 250, 500, 850)))
orig = lib.linear


def wrapped(a, w, bias, out, **kw):
    r = orig(a, w, bias, out, **kw)
    if a.dtype == torch.float32:
        torch.cuda.synchronize()
        n = kw.get("n") or w.shape[0]
        bad = not torch.isfinite(out[:, :n]).all().item()
        mode = lib._f32_state()[0]
        print(f"linear M={a.shape[0]} N={n} K={a.shape[1]} mode={mode} act={kw.get('act', 0)} max|a|={a.abs().max().item():.3g} "
              f"max|w|={w.abs().max().item():.3g} max|out|={out[:, :n].abs().max().item():.3g} {'NON-FINITE' if bad else ''}")
    return r


eng.lib.linear = wrapped
with torch.inference_mode():
    pred = model.forward(batch)
print("finite:", torch.isfinite(pred.surf_vars["2t"]).all().item())
