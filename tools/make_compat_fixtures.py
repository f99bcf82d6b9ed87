"""Fixtures for the checkpoint adapters (run in the build container; needs /root/reference).

    python tools/make_compat_fixtures.py

For each published checkpoint family (ERA5 pretrained, air pollution, ocean wave) this builds a synthetic checkpoint
in the PUBLISHED (old) key layout -- recipe below, values from oracle/detdata.py so that tests can regenerate the
inputs without storing them -- runs the REFERENCE's adapter chain on it (aurora/model/compat.py:18-284 as called by
`Aurora._adapt_checkpoint`, aurora/model/aurora.py:458-467 and the subclasses' overrides), checks that the reference
model loads the result with `strict=True` (so the recipe covers the whole schema), and stores key -> (shape, CRC-32 of the float32 bytes) of the
adapted tensors in tests/golden/compat_fixtures.json.gz (the tensors themselves would be 60 MB).
tests/test_compat.py replays the recipe through aurora_amd/model/compat.py.
"""
import gzip
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools" / "ref_stub"), "/root/reference"]

import aurora as ref  # noqa: E402

from tests.compat_recipes import FAMILIES, digest, old_layout  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def main() -> None:
    fixtures = {}
    for fam, spec in FAMILIES.items():
        model = getattr(ref, spec["cls"])(**spec["kwargs"])
        sd_new = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        old = old_layout(fam, sd_new, spec["patch"])
        adapted = model._adapt_checkpoint({k: v.clone() for k, v in old.items()})
        missing = set(sd_new) - set(adapted)
        extra = set(adapted) - set(sd_new)
        assert not missing and not extra, (fam, sorted(missing)[:5], sorted(extra)[:5])
        model.load_state_dict(adapted, strict=True)
        fixtures[fam] = digest(adapted)
        print(f"{fam}: {len(old)} published-layout entries -> {len(adapted)} adapted entries")

    # history extension (aurora.py:469-504)
    spec = FAMILIES["pretrained"]
    model = getattr(ref, spec["cls"])(**dict(spec["kwargs"], max_history_size=5))
    sd2 = {k: tuple(v.shape) for k, v in getattr(ref, spec["cls"])(**spec["kwargs"]).state_dict().items()}
    old = old_layout("pretrained", sd2, spec["patch"])
    d = model._adapt_checkpoint({k: v.clone() for k, v in old.items()})
    model.adapt_checkpoint_max_history_size(d)
    model.load_state_dict(d, strict=True)
    fixtures["history5"] = digest({k: v for k, v in d.items() if "token_embeds.weights" in k})
    print("history extension: ok")
    with gzip.open(GOLD / "compat_fixtures.json.gz", "wt") as f:
        json.dump(fixtures, f, sort_keys=True)


if __name__ == "__main__":
    main()
