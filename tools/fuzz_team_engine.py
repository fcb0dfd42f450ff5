#!/usr/bin/env python3
"""Randomised whole recursions walked by an in-process team through the level engine, against the single device
(run on an MI355X; a tool, not collected by pytest -- tests/test_gpu_team.py holds the fixed cases).

    python tools/fuzz_team_engine.py [--seconds 120] [--seed 0]

Every case draws a forest (taxa, trees, coverage, weighting, contraction), a world size (2-4) and a collective
threshold, and walks it three ways with the same seed: one device, the team with the level engine on every rank
(the level's larger nodes dealt over the ranks, those above the threshold solved collectively: levels.Engine._process)
and -- every fourth case -- the team with Team.level_engine off.  All must return the same Newick string and leave the
stream at the same draw (reference: src/sc_supertree/scs.py:158-166, one RandomState through every node).
"""
import os as _os

_os.environ.setdefault("SCS_DEBUG", "1")  # (the engine is forced onto small forests: SCS_SPEC_MIN_NODES)
_os.environ.setdefault("SCS_SPEC_MIN_NODES", "0")
import argparse
import sys
import time
import traceback
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from spectralclustersupertree_amd import levels, scs, synthetic  # noqa: E402
from spectralclustersupertree_amd.partition import LocalTeams  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    rs = np.random.RandomState(args.seed)
    scs.default_device()
    t_end = time.time() + args.seconds
    n_cases = failures = dealt = collective = mismatches = 0
    last = time.time()
    while time.time() < t_end:
        n = int(rs.choice([150, 300, 500, 800, 1300, 2100, 3000]))
        m = int(rs.choice([3, 5, 8, 13, 20, 40]))
        leaves = max(3, int(n * rs.choice([0.3, 0.6, 0.9, 1.0, 1.0])))
        strategy = str(rs.choice(["one", "depth", "branch", "branch", "bootstrap"]))
        contract = bool(rs.randint(4))
        world = int(rs.choice([2, 2, 3, 4]))
        shard_min = int(rs.choice([130, 200, 400, 900, 1 << 30]))
        seed, draw_seed = int(rs.randint(1 << 30)), int(rs.randint(1 << 30))
        what = dict(seed=seed, taxa=n, trees=m, leaves=leaves, strategy=strategy, contract=contract, world=world,
                    shard_min=shard_min, draw_seed=draw_seed)

        def make():
            return synthetic.tree_arrays(seed, n, m, leaves_per_tree=leaves, random_weights=True)

        def walk(arrays, team):
            r = np.random.RandomState(draw_seed)
            tree = scs._construct(arrays, strategy, contract, r, team=team)
            return tree.get_newick(), int(r.randint(1 << 30))

        try:
            want = walk(make(), None)
            mismatches += levels.stats["mismatches"]
            for engine in ([True, False] if n_cases % 4 == 0 else [True]):
                copies = [make() for _ in range(world)]
                teams = LocalTeams(world, shard_min=shard_min)
                for t in teams.teams:
                    t.level_engine = engine
                try:
                    out = teams.run(lambda team: walk(copies[team.rank], team))
                finally:
                    teams.close()
                if engine:
                    dealt += levels.stats["team_dealt"]
                    collective += levels.stats["team_collective"]
                for r_, got in enumerate(out):
                    if got != want:
                        failures += 1
                        print(f"FAIL {what} level_engine={engine} rank {r_}: next draw {got[1]} vs {want[1]}, "
                              f"same newick {got[0] == want[0]}", flush=True)
        except Exception:  # noqa: BLE001
            failures += 1
            print(f"ERROR {what}\n{traceback.format_exc()}", flush=True)
        n_cases += 1
        if time.time() - last > 20:
            last = time.time()
            print(f"... {n_cases} cases, {failures} failures, {dealt} nodes dealt, {collective} collective solves", flush=True)
    print(f"fuzz_team_engine seed {args.seed}: {n_cases} recursions x (one device, team with the level engine"
          f"[, team without]), {dealt} nodes dealt, {collective} collective solves (summed over ranks), "
          f"{mismatches} provisional partitions repaired on the single device: {failures} failures")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
