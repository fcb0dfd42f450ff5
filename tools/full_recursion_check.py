#!/usr/bin/env python3
"""Whole ``construct_supertree`` recursion at a BASELINE.json size, with property checks.

    python tools/full_recursion_check.py --taxa 100000 --trees 5000 --weights [--out FILE]

configs[4] ("100 000 taxa / 5 000 trees, branch + per-tree weights, full recursion") is too
large for the oracle; what can be checked on the result itself (reference behaviour:
src/sc_supertree/scs.py:96-174):
  * every taxon is a leaf of the supertree exactly once,
  * the two subtrees under the root hold exactly the taxa the top-level spectral call
    labelled 0 and 1 (label 0 first, as the reference's partition list is visited),
  * every spectral call's vertices partition that call's taxa (no taxon lost or doubled),
  * the run is deterministic: the digest of the Newick string and the RandomState's position
    after the run are printed -- equal across runs (compare two outputs of this tool).
Prints the top-level time (upload, build, solve, labels) beside the whole-recursion time.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import sys
import time
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def run(n: int, m: int, weights: bool, strategy: str = "branch", seed: int = 0, log=print) -> dict:
    from spectralclustersupertree_amd import scs, synthetic

    sys.setrecursionlimit(1_000_000)
    warnings.simplefilter("ignore")
    t0 = time.perf_counter()
    arrays = synthetic.tree_arrays(seed, n, m, random_weights=weights)
    t_gen = time.perf_counter() - t0
    log(f"input: {arrays.n_trees} trees, {len(arrays.parent)} nodes, generated in {t_gen:.1f} s")
    scs.default_device()

    calls = {"n": 0, "t": 0.0, "first_s": None, "sizes": [], "last_log": time.perf_counter()}
    real = scs.spectral_bipartition_device

    def timed(tables, rs, **kw):
        t1 = time.perf_counter()
        out = real(tables, rs, **kw)
        dt = time.perf_counter() - t1
        if calls["first_s"] is None:
            calls["first_s"] = dt
        calls["n"] += 1
        calls["t"] += dt
        calls["sizes"].append(tables.n_taxa)
        now = time.perf_counter()
        if now - calls["last_log"] > 30.0:
            calls["last_log"] = now
            log(f"  ... {calls['n']} spectral calls, {now - t_start:.0f} s")
        return out

    scs.spectral_bipartition_device = timed
    rs = np.random.RandomState(seed)
    t_start = time.perf_counter()
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with scs.trace_nodes() as trace:
                tree = scs._construct(arrays, strategy, True, rs)
        trace = list(trace)
    finally:
        scs.spectral_bipartition_device = real
    # (scs._fiedler_checked clusters a block whose residual stopped between tol and ACCEPT_RESIDUAL with a
    # RuntimeWarning: a place where labels could leave parity silently -- counted, expected never)
    accepted = sum(1 for w in caught if issubclass(w.category, RuntimeWarning) and "Fiedler solve stopped" in str(w.message))
    t_total = time.perf_counter() - t_start
    log(f"recursion done: {t_total:.1f} s, {calls['n']} spectral calls")

    # ---- properties
    tips = tree.get_tip_names()
    all_names = {arrays.name(i) for i in range(arrays.n_taxa)}
    every_taxon_once = len(tips) == arrays.n_taxa and set(tips) == all_names
    top = trace[0]
    side = [set(), set()]
    for vertex, lab in zip(top["vertices"], top["labels"]):
        side[int(lab)].update(vertex)
    kids = list(tree.children)
    top_parts_match = (len(kids) == 2 and set(kids[0].get_tip_names()) == side[0]
                       and set(kids[1].get_tip_names()) == side[1])
    partitions_ok = True
    for entry in trace:
        seen = [x for v in entry["vertices"] for x in v]
        if len(seen) != len(set(seen)):
            partitions_ok = False
    newick = tree.get_newick()
    sizes = np.asarray([len(e["labels"]) for e in trace] if trace else calls["sizes"])
    from spectralclustersupertree_amd import levels

    engine = dict(levels.stats)
    engine["mismatch_sizes"] = sorted(engine["mismatch_sizes"], reverse=True)[:20]
    big = engine.pop("big_jobs", [])
    engine["solves_of_4096_vertices_and_more"] = {"count": len(big), "seconds": round(sum(t for _, t in big), 2),
                                                  "largest": sorted(big, reverse=True)[:12]}
    splits = engine.pop("split_log", [])
    engine["level_splits_of_5_ms_and_more"] = {"count": len(splits), "seconds": round(sum(t for _, t in splits), 2),
                                                    "largest": sorted(splits, reverse=True)[:12]}
    redo = engine.pop("redo_log", [])
    engine["redo_by_vertices"] = {
        f"{lo + 1}-{hi}": {"nodes": len(sel), "seconds": round(sum(r[1] for r in sel), 3), "levels": sum(r[2] for r in sel)}
        for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 512), (512, 2048), (2048, 1 << 30))
        if (sel := [r for r in redo if lo < r[0] <= hi])}
    for key in list(engine):
        if isinstance(engine[key], float):
            engine[key] = round(engine[key], 3)
    return {
        "n_taxa": n, "n_trees": m, "per_tree_weights": weights, "pcg_weighting": strategy, "seed": seed,
        "generate_s": round(t_gen, 2),
        "whole_recursion_s": round(t_total, 2),
        "top_level_s": round(calls["first_s"], 3),
        "top_level_what": "spectral_bipartition_device at the root: contraction groups, tables upload, "
                          "scs_pcg_build, scs_fiedler, k_means",
        "spectral_calls": int(len(trace)),
        "spectral_calls_on_the_node_by_node_path": int(calls["n"]),
        "level_engine": engine,
        "accepted_residual_warnings": int(accepted),
        "level_engine_max_taxa": levels.max_taxa(),
        "in_spectral_calls_s": round(calls["t"], 2),
        "largest_problems": sorted(sizes.tolist(), reverse=True)[:6],
        "calls_by_size": {"<=64": int(np.sum(sizes <= 64)), "65-512": int(np.sum((sizes > 64) & (sizes <= 512))),
                          "513-4096": int(np.sum((sizes > 512) & (sizes <= 4096))), ">4096": int(np.sum(sizes > 4096))},
        "calls_by_size_fine": {f"{lo}-{hi}": int(np.sum((sizes >= lo) & (sizes <= hi)))
                               for lo, hi in ((2, 4), (5, 8), (9, 16), (17, 24), (25, 32), (33, 64), (65, 96), (97, 128),
                                              (129, 256), (257, 512), (513, 1024), (1025, 4096))},
        "every_taxon_exactly_once": bool(every_taxon_once),
        "top_level_parts_equal_top_level_labels": bool(top_parts_match),
        "every_call_partitions_its_taxa": bool(partitions_ok),
        "top_level_split": [len(side[0]), len(side[1])],
        "newick_sha256": hashlib.sha256(newick.encode()).hexdigest(),
        "random_state_next_draw": int(rs.randint(1 << 30)),
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=100000)
    ap.add_argument("--trees", type=int, default=5000)
    ap.add_argument("--weights", action="store_true")
    ap.add_argument("--strategy", default="branch")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    res = run(args.taxa, args.trees, args.weights, args.strategy, args.seed,
              log=lambda s: print(s, flush=True))
    text = json.dumps(res, indent=1)
    print(text, flush=True)
    if args.out:
        Path(args.out).write_text(text + "\n")
    ok = (res["every_taxon_exactly_once"] and res["top_level_parts_equal_top_level_labels"]
          and res["every_call_partitions_its_taxa"])
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
