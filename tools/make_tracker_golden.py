"""Run the REFERENCE cyclone tracker (aurora/tracker.py, imported from /root/reference through tools/ref_stub) on the
synthetic storm of tests/tracker_scenario.py and store its track as tests/golden/tracker_track.json.

    python tools/make_tracker_golden.py
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools" / "ref_stub"), "/root/reference"]

import aurora as ref  # noqa: E402

from tests import tracker_scenario as sc  # noqa: E402

out = {}
for name, lon0 in sc.SCENARIOS.items():
    tracker = ref.Tracker(init_lat=sc.START[0], init_lon=lon0, init_time=sc.START[2])
    for step in range(1, sc.STEPS + 1):
        tracker.step(sc.batch(step, ref.Batch, ref.Metadata, lon0=lon0))
    df = tracker.results()
    out[name] = {"fails": tracker.fails, "time": [t.isoformat() for t in df["time"]],
                 **{k: [None if x != x else float(x) for x in df[k]] for k in ("lat", "lon", "msl", "wind")}}
    print(name, "fails:", tracker.fails)
    print(df)
path = ROOT / "tests" / "golden" / "tracker_track.json"
path.write_text(json.dumps(out, indent=1))
print("->", path)
