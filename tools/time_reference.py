"""Time the REAL reference (`/root/reference`, imported through tools/ref_stub) and the oracle port on this machine's
host cores, on the headline workload (AuroraPretrained fp32, 721 x 1440 x 13, one forward) or a sub-grid of it.

    python tools/time_reference.py [--grid 721x1440] [--threads N] [--out profiles/r02_reference_cpu.json]

This is the recipe behind BASELINE.md section 2 (201 s per step on 8 Xeon cores) and behind the relation between the
reference and the oracle port that `bench.py`'s `cpu_baseline` leg quotes: /root/reference does not exist on the GPU
box, so bench.py can only time the port there; this script shows, where both exist, how the two compare on identical
inputs.  Weights are seeded random (timing does not depend on their values), inputs as in bench.py.
"""
import argparse
import json
import os
import platform
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools" / "ref_stub"), "/root/reference"]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="721x1440")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r02_reference_cpu.json"))
    ap.add_argument("--skip-port", action="store_true")
    args = ap.parse_args()
    H, W = map(int, args.grid.split("x"))
    torch.set_num_threads(args.threads)

    import aurora as ref                       # the reference itself
    from aurora import normalisation as ref_nz

    import aurora_amd
    from bench import LEVELS, synthetic_batch
    from oracle import aurora_oracle as oracle

    with torch.device("meta"):
        cfg = aurora_amd.AuroraPretrained().config
    t0 = time.perf_counter()
    torch.manual_seed(0)
    model = ref.AuroraPretrained(autocast=False).eval()
    with torch.no_grad():
        for p in model.parameters():
            if not p.any():
                p.normal_(std=0.02)
    print(f"reference model built in {time.perf_counter() - t0:.1f} s", flush=True)
    b = synthetic_batch(cfg, H, W, 1, "cpu")
    batch = ref.Batch(b.surf_vars, b.static_vars, b.atmos_vars,
                      ref.Metadata(b.metadata.lat, b.metadata.lon, b.metadata.time, LEVELS))
    res = {"grid": [H, W], "threads": args.threads, "cpu": platform.processor() or platform.machine(),
           "torch": torch.__version__, "model": "AuroraPretrained fp32, 1.3B parameters, T=2, 13 levels"}
    with torch.inference_mode():
        t0 = time.perf_counter()
        pred = model.forward(batch)
        res["reference_s_per_step"] = time.perf_counter() - t0
    print(f"reference forward: {res['reference_s_per_step']:.1f} s", flush=True)
    if not args.skip_port:
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        with torch.inference_mode():
            t0 = time.perf_counter()
            o_s, o_a, _ = oracle.forward(sd, cfg, b.surf_vars, b.static_vars, b.atmos_vars, b.metadata.lat,
                                         b.metadata.lon, b.metadata.time, LEVELS, 0, ref_nz.locations, ref_nz.scales)
            res["port_s_per_step"] = time.perf_counter() - t0
        print(f"oracle port forward: {res['port_s_per_step']:.1f} s", flush=True)
        worst = 0.0
        for k, v in pred.surf_vars.items():
            worst = max(worst, ((v - o_s[k]).abs().mean() / o_s[k].abs().mean()).item())
        for k, v in pred.atmos_vars.items():
            worst = max(worst, ((v - o_a[k]).abs().mean() / o_a[k].abs().mean()).item())
        res["port_vs_reference_mean_rel_err"] = worst
        res["port_over_reference_time"] = res["port_s_per_step"] / res["reference_s_per_step"]
    res["reference_steps_per_s"] = 1.0 / res["reference_s_per_step"]
    Path(args.out).write_text(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
