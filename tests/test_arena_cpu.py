"""The device-memory arena's logic (csrc/scs_arena.h) on the CPU: tests/native/arena_harness.cpp drives the
same header the HIP library compiles, with malloc-like backing under a budget and scripted streams.

Checked there: chunks in use never overlap, every byte of every slab is in exactly one chunk and the indices
are complete, memory released by one context is never handed to ANOTHER before the work its owner had queued
at the release is done (its owner may take it at once), the driver's refusal makes the arena wait / hand slabs
back / try again, and at the end everything is back with the driver.
"""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("arena") / "arena_harness"
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-o", str(out),
                    str(ROOT / "tests" / "native" / "arena_harness.cpp")], check=True)
    return out


def test_arena_scenarios(harness):
    res = subprocess.run([str(harness), "scenario"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    for what in ("carve/coalesce/trim", "small/large slabs", "pending chunks", "markers", "driver refusal", "owner_gone"):
        assert f"ok {what}" in res.stdout


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_arena_random_requests(harness, seed):
    res = subprocess.run([str(harness), "random", str(seed), "20000"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok random")
