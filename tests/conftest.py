"""pytest configuration: markers and import paths."""

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

sys.setrecursionlimit(10000)

# The tests hold the product's A/B paths against each other (SCS_SPLIT_SMALL, SCS_TREE_PARALLEL, SCS_SPEC_VOTES,
# ...): probe switches, which the libraries and the package only look at when SCS_DEBUG=1 was in the environment
# at load time (csrc/scs_internal.h scs_dbg, spectralclustersupertree_amd/_env.py).  Set before anything loads.
import os  # noqa: E402

os.environ.setdefault("SCS_DEBUG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes on the GPU box (configs[4], 80 GB of W); "
                                       "deselect with -m 'gpu and not slow'")


def _gpu_available() -> bool:
    try:
        from spectralclustersupertree_amd import _native

        return _native.load_library().scs_device_count() > 0
    except (ImportError, OSError, AttributeError):
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a device (or without the built library) skips instead of
    erroring; the product path itself still fails loudly (tests/test_boundary_cpu.py)."""
    import pytest

    if not any("gpu" in item.keywords for item in items):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device / libscs_hip.so not built")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
