"""aurora_amd: an MI355X-native forward / rollout engine for the Aurora model family.

Public surface = the reference's (aurora/__init__.py:3-29).  `Tracker` is exported as a stub: the cyclone tracker
(aurora/tracker.py) is CPU post-processing of finished predictions, outside the forward / roll-out hot path this
package accelerates.
"""

from aurora_amd.batch import Batch, Metadata
from aurora_amd.model.aurora import (
    Aurora,
    Aurora12hPretrained,
    AuroraAirPollution,
    AuroraHighRes,
    AuroraPretrained,
    AuroraSmall,
    AuroraSmallPretrained,
    AuroraWave,
)
from aurora_amd.rollout import rollout, write_rollout


class Tracker:
    """Placeholder for the reference's tropical-cyclone tracker (aurora/tracker.py).

    The tracker consumes finished predictions on the CPU (scipy / numpy peak finding); nothing of it is on the
    forward / roll-out path.  `Batch` objects produced by aurora_amd have exactly the reference's fields, so the
    reference's own `Tracker` can be fed with them: `aurora.Tracker(...).step(pred.to("cpu"))`."""

    def __init__(self, *args, **kwargs) -> None:
        raise NotImplementedError(
            "aurora_amd accelerates Aurora.forward / rollout only; use the reference's `aurora.Tracker` on the "
            "predictions (they are field-compatible `Batch` objects)."
        )

__all__ = [
    "Aurora",
    "AuroraPretrained",
    "AuroraSmallPretrained",
    "AuroraSmall",
    "Aurora12hPretrained",
    "AuroraHighRes",
    "AuroraAirPollution",
    "AuroraWave",
    "Batch",
    "Metadata",
    "rollout",
    "write_rollout",
    "Tracker",
]
