"""aurora_amd: an MI355X-native forward / rollout engine for the Aurora model family.

Public surface = the reference's (aurora/__init__.py:3-29), minus the CPU-side cyclone
`Tracker` (out of the hot-path scope).
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
from aurora_amd.rollout import rollout

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
]
