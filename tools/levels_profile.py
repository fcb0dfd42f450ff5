#!/usr/bin/env python3
"""cProfile of one whole recursion with the level-synchronous engine (tools/levels_check.py's input shapes).

    python tools/levels_profile.py TAXA TREES [STRATEGY] [--no-profile]
"""
import cProfile
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import pstats
import sys
import time
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    n, m = int(sys.argv[1]), int(sys.argv[2])
    strategy = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "branch"
    from spectralclustersupertree_amd import levels, scs, synthetic

    sys.setrecursionlimit(1_000_000)
    warnings.simplefilter("ignore")
    arrays = synthetic.tree_arrays(3, n, m, random_weights=True)
    scs.default_device()
    # warm-up on a small input (library self-tests, contexts)
    scs._construct(synthetic.tree_arrays(5, 300, 30), strategy, True, np.random.RandomState(0))
    rs = np.random.RandomState(0)
    prof = None if "--no-profile" in sys.argv else cProfile.Profile()
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    tree = scs._construct(arrays, strategy, True, rs)
    if prof:
        prof.disable()
    dt = time.perf_counter() - t0
    st = dict(levels.stats)
    st["mismatch_sizes"] = sorted(st["mismatch_sizes"], reverse=True)[:12]
    big = st.pop("big_jobs", [])
    splits = st.pop("split_log", [])
    print("level splits of 5 ms and more (leaves, s):", len(splits), "splits,", round(sum(t for _, t in splits), 2), "s in all;",
          sorted(splits, reverse=True)[:16])
    redo = st.pop("redo_log", [])
    for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 512), (512, 2048), (2048, 1 << 30)):
        sel = [r for r in redo if lo < r[0] <= hi]
        if sel:
            print(f"redo of nodes of {lo + 1}..{hi} vertices: {len(sel)} nodes, {sum(r[1] for r in sel):.3f} s, "
                  f"{sum(r[2] for r in sel)} levels")
    print("solves of 4 096 vertices and more:", len(big), "jobs,", round(sum(t for _, t in big), 2), "s in all;", sorted(big, reverse=True)[:40])
    print(f"{n} taxa / {m} trees {strategy}: {dt:.2f} s; next draw {rs.randint(1 << 30)}; {st}")
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(45)
        pstats.Stats(prof).sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
