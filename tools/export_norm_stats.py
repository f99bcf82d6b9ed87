"""Export the reference's per-variable normalisation statistics to a data file.

The location/scale tables (aurora/normalisation.py:77-457 in the reference) are
dataset statistics that published checkpoints were trained against; a drop-in
engine must use the very same numbers.  They are data, so they are exported
once into aurora_amd/data/norm_stats.json by this script (run in the build
container, where /root/reference exists) rather than retyped.

    python tools/export_norm_stats.py
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "tools" / "ref_stub"), "/root/reference"]

from aurora import normalisation as ref_norm  # noqa: E402

out = {
    "locations": {k: float(v) for k, v in ref_norm.locations.items()},
    "scales": {k: float(v) for k, v in ref_norm.scales.items()},
}
dst = ROOT / "aurora_amd" / "data" / "norm_stats.json"
dst.write_text(json.dumps(out, indent=0, sort_keys=True) + "\n")
print(f"wrote {dst}: {len(out['locations'])} locations, {len(out['scales'])} scales")
