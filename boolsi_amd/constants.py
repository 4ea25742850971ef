"""
Shared enums and names of the BoolSi-compatible front end.

Mirrors the values of the reference's `boolsi/constants.py:11-48` (Mode, NodeStateRange,
result names, log date format) because they are part of the YAML / CLI surface and of the
problem enumeration (`boolsi/batching.py:171-175`).  `MpiTags` has no counterpart: the task
farm is replaced by range partitioning.
"""
from enum import Enum
from math import inf  # noqa: F401  (re-exported, the reference's callers pass `inf` for unset caps)


class Mode(Enum):
    SIMULATE = 0
    ATTRACT = 1
    TARGET = 2


mode_descriptions = {Mode.SIMULATE: 'simulate', Mode.ATTRACT: 'attract', Mode.TARGET: 'target'}


class NodeStateRange(Enum):
    """Which states a varied fixed node / perturbation may take (value codes as in the reference)."""
    MAYBE_FALSE = 0            # '0?'   -> {absent, 0}
    MAYBE_TRUE = -1            # '1?'   -> {absent, 1}
    MAYBE_TRUE_OR_FALSE = -10  # 'any?' -> {absent, 0, 1}
    TRUE_OR_FALSE = 10         # 'any'  -> {0, 1}


# Engine-side codes of NodeStateRange (include/bsx.h: BSX_RANGE_*), chosen so that
# digit d of a variation maps to:  code 0: {absent, 0}   code 1: {absent, 1}
#                                  code 2: {0, 1}        code 3: {absent, 0, 1}
RANGE_CODE = {
    NodeStateRange.MAYBE_FALSE: 0,
    NodeStateRange.MAYBE_TRUE: 1,
    NodeStateRange.TRUE_OR_FALSE: 2,
    NodeStateRange.MAYBE_TRUE_OR_FALSE: 3,
}
RANGE_RADIX = {0: 2, 1: 2, 2: 2, 3: 3}

simulation_name = 'simulation'
aggregated_attractor_name = 'attractor'

log_date_format = '%d-%b-%Y %H:%M:%S'
