#!/usr/bin/env python3
"""Randomised whole-recursion comparison with the oracle (run on an MI355X; a tool, not collected
by pytest -- tests/test_gpu_recursion.py holds the fixed cases).

    python tests/fuzz_recursion.py [--seconds 120] [--seed 0]

Every case draws a forest (taxa, trees, coverage, twins that force contraction, polytomies by
zero-length splicing of the input are not needed: partial coverage already yields component
splits and dropped trees), a weighting, tree weights and the contraction flag, and runs
``construct_supertree`` (tree objects or flat arrays) against ``oracle/scs_oracle`` node by
node: the same vertices and the IDENTICAL label vector at every spectral call, the same
topology, the RandomState left in the same state.  Forests of a few trees tie exactly (taxa of one
clade are equivalent towards the rest); a tie must be PROVEN
(tests/test_gpu_recursion.compare_with_oracle) to be followed.
"""
import os as _os

_os.environ.setdefault("SCS_DEBUG", "1")  # (the sweeps force probe paths: csrc/scs_internal.h scs_dbg)
import argparse
import sys
import time
import traceback
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from test_gpu_recursion import TIE_PROOFS, compare_with_oracle, recursion_input  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-taxa", type=int, default=260)
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    rs = np.random.RandomState(args.seed)
    t_end = time.time() + args.seconds
    n_cases = n_calls = n_ties = failures = 0
    by_strategy = {}
    last = time.time()
    while time.time() < t_end:
        n_taxa = int(rs.choice([5, 8, 13, 21, 34, 55, 70, 100, 160, args.max_taxa]))
        n_trees = int(rs.choice([2, 3, 5, 8, 13, 20]))
        leaves = max(3, int(n_taxa * rs.choice([0.4, 0.6, 0.8, 1.0])))
        n_twins = int(rs.choice([0, 0, 1, 2, max(1, n_taxa // 10)]))
        n_twins = min(n_twins, n_taxa)
        strategy = str(rs.choice(["one", "depth", "branch", "branch", "bootstrap"]))
        weighted = bool(rs.randint(2))
        contract = bool(rs.randint(4))  # mostly on, as the reference's default
        as_arrays = bool(rs.randint(2))
        case_seed = int(rs.randint(1 << 30))
        what = dict(seed=case_seed, taxa=n_taxa, trees=n_trees, leaves=leaves, twins=n_twins, strategy=strategy,
                    weighted=weighted, contract=contract, arrays=as_arrays)
        try:
            trees, weights = recursion_input(case_seed % 100000, n_taxa, n_trees, leaves, n_twins, weighted)
            trace, ties = compare_with_oracle(trees, weights, strategy, seed=case_seed % 9973,
                                              contract_edges=contract, as_arrays=as_arrays,
                                              ties_allowed=True)
            n_calls += len(trace)
            n_ties += len(ties)
            by_strategy[strategy] = by_strategy.get(strategy, 0) + 1
        except ValueError as exc:
            # an input no source tree covers (the reference raises the same way): both sides raise --
            # compare_with_oracle lets the product's exception through first
            if "at least one tree" not in str(exc):
                failures += 1
                print("FAIL", what, "->", repr(exc), flush=True)
        except Exception as exc:  # noqa: BLE001
            failures += 1
            print("FAIL", what, "->", repr(exc), flush=True)
            traceback.print_exc()
        n_cases += 1
        if time.time() - last > 30:
            last = time.time()
            print(f"... {n_cases} cases, {n_calls} spectral calls, {n_ties} proven ties, {failures} failures",
                  flush=True)
    print(f"fuzz_recursion: {n_cases} recursions ({by_strategy}), {n_calls} spectral calls compared node by node, "
          f"{n_ties} proven ties followed ({dict(TIE_PROOFS)}), {failures} failures")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
