"""aurora_amd: an MI355X-native forward / rollout engine for the Aurora model family.

Public surface = the reference's (aurora/__init__.py:3-29).  The cyclone tracker (aurora/tracker.py) follows a roll-out
without moving the predictions off the device: only its search windows travel (aurora_amd/tracker.py).
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
from aurora_amd.tracker import Tracker

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
