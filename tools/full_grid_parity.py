"""Full-grid parity of the other BASELINE.json configurations against the CPU oracle (bench.py does this for configs[1]):
the GPU step (bf16 backbone, and the fp32 engine on the same weights) vs `oracle.forward` on the same seeded weights and
synthetic Batch, on the configuration's own grid.  Run on the GPU box; one JSON line per case.

    python tools/full_grid_parity.py [highres] [airpollution]

  highres       AuroraHighRes (patch 10, LoRA), 1801 x 3600 -- BASELINE configs[3]'s grid on one GPU
  airpollution  AuroraAirPollution (CAMS, patch 3, level-conditioned, second decoder Perceiver), 451 x 900 -- configs[4]
"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

CASES = {"highres": ("AuroraHighRes", 1801, 3600), "airpollution": ("AuroraAirPollution", 451, 900)}

for case in sys.argv[1:] or list(CASES):
    cls_name, H, W = CASES[case]
    model = bench.build_model("cuda", cls_name)
    batch = bench.synthetic_batch(model.config, H, W, 1, "cuda")
    with torch.inference_mode():
        model.forward(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred = model.forward(batch)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    cpu, oracle_out = bench.cpu_baseline(model, H, W, 1500.0, cls_name)
    out = {"case": f"{cls_name} {H}x{W}", "gpu_ms_per_step": ms, "cpu_oracle": cpu}
    if oracle_out is not None:
        out["parity_full_grid"] = bench.full_grid_parity(model, batch, pred, oracle_out)
    print(json.dumps(out), flush=True)
    del model, batch, pred
    torch.cuda.empty_cache()
