"""Generate golden vectors from the upstream reference (run in the build container).

    python tools/make_golden.py [case ...]

For every case in tests/golden_cases.py this script builds the REFERENCE model
(/root/reference, imported through tools/ref_stub for the missing `timm`), loads the
deterministic weights of oracle/detdata.py, runs `rollout()` in fp64 on deterministic
inputs, and stores the predictions as tests/golden/<case>.npz.  It also runs the oracle
on the same data and prints the deviation, and it (re)writes the state_dict schema
fixture tests/golden/state_dict_schemas.json.gz.
"""
import gzip
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools" / "ref_stub"), "/root/reference"]

import aurora as ref  # noqa: E402
from aurora import normalisation as ref_norm  # noqa: E402

import aurora_amd  # noqa: E402
from oracle import aurora_oracle as oracle  # noqa: E402
from oracle import detdata  # noqa: E402
from tests.golden_cases import CASES  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def run_case(name: str, case: dict) -> None:
    torch.manual_seed(0)
    model = getattr(ref, case["cls"])(**case["kwargs"]).double().eval()
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    sd = detdata.det_state_dict(shapes, torch.float64)
    model.load_state_dict(sd, strict=True)

    with torch.device("meta"):
        mine = getattr(aurora_amd, case["cls"])(**case["kwargs"])
    cfg = mine.config
    if case["cls"] == "AuroraWave":
        surf, static, atmos, lat, lon, times = detdata.det_wave_inputs(
            cfg.static_vars, cfg.atmos_vars, case["B"], case["T"], case["H"], case["W"], case["levels"],
            ref_norm.locations, ref_norm.scales)
    else:
        surf, static, atmos, lat, lon, times = detdata.det_inputs(
            cfg.surf_vars, cfg.static_vars, cfg.atmos_vars, case["B"], case["T"], case["H"], case["W"],
            case["levels"], ref_norm.locations, ref_norm.scales,
            positive=cfg.positive_surf_vars + cfg.positive_atmos_vars,
        )
    batch = ref.Batch(surf, static, atmos, ref.Metadata(lat, lon, times, tuple(case["levels"])))
    out = {}
    with torch.inference_mode():
        preds = list(ref.rollout(model, batch, steps=case["steps"]))
        gen = oracle.rollout(sd, cfg, surf, static, atmos, lat, lon, times, case["levels"],
                             case["steps"], ref_norm.locations, ref_norm.scales,
                             variant=mine.variant)
        worst = 0.0
        for s, (pred, (osurf, oatmos, otimes)) in enumerate(zip(preds, gen)):
            assert pred.metadata.rollout_step == s + 1 and pred.metadata.time == otimes
            for kind, rd, od in (("surf", pred.surf_vars, osurf), ("atmos", pred.atmos_vars, oatmos)):
                assert tuple(rd) == tuple(od), (tuple(rd), tuple(od))
                for k, v in rd.items():
                    out[f"s{s}.{kind}.{k}"] = v.numpy().astype(np.float32)
                    assert torch.equal(torch.isnan(v), torch.isnan(od[k])), k
                    err = ((v - od[k]).abs().nan_to_num(0).max().item()
                           / (v.abs().nan_to_num(0).max().item() + 1e-30))
                    worst = max(worst, err)
    np.savez(GOLD / f"{name}.npz", **out)
    size = (GOLD / f"{name}.npz").stat().st_size / 1e6
    print(f"{name:16s} {len(out):3d} arrays {size:6.2f} MB   oracle-vs-reference max rel err {worst:.2e}")


def write_schemas() -> None:
    """state_dict key -> shape for every public class (default constructor arguments)."""
    from torch.nn import init

    noop = lambda t, *a, **k: t  # noqa: E731  (skip the expensive initialisers)
    init.trunc_normal_ = noop
    import aurora.model.util as ref_util

    ref_util.trunc_normal_ = noop
    schemas = {}
    for cls in ("Aurora", "AuroraPretrained", "AuroraSmallPretrained", "Aurora12hPretrained",
                "AuroraHighRes", "AuroraAirPollution", "AuroraWave"):
        m = getattr(ref, cls)()
        schemas[cls] = {k: list(v.shape) for k, v in m.state_dict().items()}
        del m
    with gzip.open(GOLD / "state_dict_schemas.json.gz", "wt") as f:
        json.dump(schemas, f, sort_keys=True)
    print("schemas:", {k: len(v) for k, v in schemas.items()})


if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        if n == "schemas":
            write_schemas()
        else:
            run_case(n, CASES[n])
    if not sys.argv[1:]:
        write_schemas()
