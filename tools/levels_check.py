#!/usr/bin/env python3
"""The level-synchronous recursion (``spectralclustersupertree_amd/levels.py``) against the node-by-node
walk on the same inputs, in ONE process: Newick string, RandomState position and the spectral-call trace
(vertices, labels) must be identical.

    python tools/levels_check.py [--cases N] [--seed S] [--big]

Not an oracle comparison (``tests/test_gpu_recursion.py`` / ``tests/fuzz_recursion.py`` hold the product
against the oracle with the engine on): this is the quick A/B for the engine's own plumbing, with timing.
"""

from __future__ import annotations

import argparse
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import sys
import time
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def run(arrays, strategy, contract, seed, spec: bool):
    from spectralclustersupertree_amd import levels, scs

    os.environ["SCS_SPEC_MAX_TAXA"] = os.environ.get("LEVELS_MAX_TAXA", "2048") if spec else "0"
    os.environ["SCS_SPEC_MIN_NODES"] = "0"
    rs = np.random.RandomState(seed)
    t0 = time.perf_counter()
    with scs.trace_nodes() as trace:
        tree = scs._construct(arrays, strategy, contract, rs)
    dt = time.perf_counter() - t0
    trace = list(trace)
    return tree.get_newick(), int(rs.randint(1 << 30)), trace, dt, dict(levels.stats)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true")
    args = ap.parse_args()
    from spectralclustersupertree_amd import synthetic

    sys.setrecursionlimit(1_000_000)
    warnings.simplefilter("ignore")
    rng = np.random.RandomState(args.seed)
    bad = 0
    shapes = [(40, 12, None), (120, 30, 80), (300, 40, None), (700, 60, 400), (1500, 40, None), (2500, 100, None)]
    if args.big:
        shapes = [(20000, 1000, None), (20000, 5000, None)]
    for case in range(args.cases):
        n, m, leaves = shapes[case % len(shapes)]
        strategy = ("branch", "depth", "one", "bootstrap")[case % 4]
        contract = case % 3 != 2
        seed = int(rng.randint(1 << 20))
        kw = {} if leaves is None else {"leaves_per_tree": leaves}
        arrays = synthetic.tree_arrays(seed, n, m, random_weights=bool(case % 2), **kw)
        a = run(arrays, strategy, contract, seed, True)
        b = run(arrays, strategy, contract, seed, False)
        same = a[0] == b[0] and a[1] == b[1] and len(a[2]) == len(b[2])
        if same:
            for x, y in zip(a[2], b[2]):
                if x["vertices"] != y["vertices"] or not np.array_equal(x["labels"], y["labels"]):
                    same = False
                    break
        st = a[4]
        print(f"case {case}: {n} taxa / {m} trees {strategy} contract={contract} seed={seed}: "
              f"{'same' if same else 'DIFFERENT'}  engine {a[3]:.2f} s, node by node {b[3]:.2f} s; "
              f"{len(a[2])} spectral calls; roots {st['roots']} levels {st['levels']} nodes {st['nodes']} "
              f"mismatches {st['mismatches']} {st['mismatch_sizes'][:8]} fallbacks {st['fallbacks']} "
              f"exact-group nodes {st['exact_group_nodes']}\n      first {st['t_first']:.2f} host {st['t_host']:.2f} small {st['t_small']:.2f} "
              f"large {st['t_large']:.2f} ({st['n_large']}) labels {st['t_labels']:.2f} split {st['t_split']:.2f} "
              f"build(walk) {st['t_build']:.2f}", flush=True)
        bad += not same
    print("FAILED" if bad else "all cases identical")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
