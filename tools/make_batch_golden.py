"""Golden values of the reference's `Batch` host methods (aurora/batch.py): `regrid`, `normalise` / `unnormalise`, `crop`
on a seeded random batch -> tests/golden/batch_methods.npz.  Run in the build container (needs /root/reference).

    python tools/make_batch_golden.py
"""
import sys
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tools" / "ref_stub"), "/root/reference"]

import aurora as ref  # noqa: E402

from tests.test_batch import seeded_batch  # noqa: E402

b = seeded_batch(ref.Batch, ref.Metadata)
out = {}
rg = b.regrid(7.5)
for grp, d in (("surf", rg.surf_vars), ("static", rg.static_vars), ("atmos", rg.atmos_vars)):
    for k, v in d.items():
        out[f"regrid.{grp}.{k}"] = v.numpy()
out["regrid.lat"], out["regrid.lon"] = rg.metadata.lat.numpy(), rg.metadata.lon.numpy()
nb = b.normalise(surf_stats={"2t": (270.0, 30.0)})
for grp, d in (("surf", nb.surf_vars), ("static", nb.static_vars), ("atmos", nb.atmos_vars)):
    for k, v in d.items():
        out[f"normalise.{grp}.{k}"] = v.numpy()
un = nb.unnormalise(surf_stats={"2t": (270.0, 30.0)})
for k, v in un.surf_vars.items():
    out[f"unnormalise.surf.{k}"] = v.numpy()
cr = b.crop(4)
out["crop.lat"] = cr.metadata.lat.numpy()
out["crop.2t"] = cr.surf_vars["2t"].numpy()
np.savez_compressed(ROOT / "tests" / "golden" / "batch_methods.npz", **out)
print({k: v.shape for k, v in out.items()})
