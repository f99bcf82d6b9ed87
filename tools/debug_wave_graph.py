"""Debug: eager vs eager vs graphed roll-out of the wave golden case, per step."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from aurora_amd import rollout
from tests.test_gpu_model import build

case, model, batch = build(sys.argv[1] if len(sys.argv) > 1 else "wave")
with torch.inference_mode():
    e1 = list(rollout(model, batch, steps=3))
    e2 = list(rollout(model, batch, steps=3))
    g = list(rollout(model, batch, steps=3, graph=True))
    g2 = list(rollout(model, batch, steps=3, graph=True))
for s in range(3):
    for nm, a, b in (("e1-e2", e1, e2), ("e1-g", e1, g), ("g-g2", g, g2)):
        worst = 0.0
        for k in a[s].surf_vars:
            x, y = a[s].surf_vars[k].nan_to_num(-1.), b[s].surf_vars[k].nan_to_num(-1.)
            worst = max(worst, ((x - y).abs().max() / y.abs().max()).item())
        print(s, nm, worst)
