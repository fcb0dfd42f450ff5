"""Environment switches of the host side, in two kinds (DESIGN.md section 12 has the table).

SUPPORTED switches select a documented behaviour and are read where they apply: ``SCS_DEVICE``, ``SCS_TEAM``,
``SCS_MULTI_MODE``, ``SCS_SHARD_MIN_VERTICES``, ``SCS_CHILD_RNG``, ``SCS_AHEAD``, ``SCS_AHEAD_WORKERS``,
``SCS_SPEC_MAX_TAXA``, ``SCS_KMEANS``, ``SCS_MALLOC_TUNE``, ``SCS_RDZV_PORT`` (and, read by the libraries,
``SCS_LOWP``, ``SCS_LOWP_MAX_BYTES``, ``SCS_WS_LIMIT_MB``, ``SCS_HOST_THREADS``).

PROBE switches -- A/B paths of committed measurements, thresholds, test hooks -- exist for ``tools/`` and
``tests/`` only and are read through ``probe``, which answers "unset" unless ``SCS_DEBUG=1`` is in the environment
(``libscs_hip.so`` applies the same gate to its own probes, ``csrc/scs_internal.h``): in a production process no
probe path can be reached, whatever else is exported."""

from __future__ import annotations

import os


def debug() -> bool:
    return bool(int(os.environ.get("SCS_DEBUG", "0") or 0))


def probe(name: str, default: str) -> str:
    """Value of the probe switch ``name``; ``default`` unless SCS_DEBUG=1 and the switch is set (and not empty)."""
    if not debug():
        return default
    return os.environ.get(name, default) or default
