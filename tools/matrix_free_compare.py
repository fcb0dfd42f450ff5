#!/usr/bin/env python3
"""Measured comparison (round 5, not the product path): the Fiedler pair from the MATRIX-FREE operator --
W applied straight from the flattened tables, scs_pcg_build never run -- beside the dense path
(scs_pcg_build + the SYMM stream) on the same tables.

    python tools/matrix_free_compare.py [--configs cfg2,cfg3,cfg4] [--out FILE]

Per configuration: wall time of (build + solve) dense and of (graph set-up + solve) matrix-free at block
widths 4 and 8, the time of one operator application of either kind, iterations, and the distance of the
matrix-free pair to the dense one (eigenvalue, embedding column, unit-norm eigenvector).  One JSON document.
"""
from __future__ import annotations

import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

CONFIGS = {  # BASELINE.json configs[2..4] (taxa, trees, per-tree weights)
    "cfg1": (1000, 100, False),
    "cfg2": (10000, 500, False),
    "cfg3": (50000, 2000, False),
    "cfg4": (100000, 5000, True),
}


def timed(fn):
    t0 = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg2,cfg3")
    ap.add_argument("--out", default="")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    from spectralclustersupertree_amd import _native as nv
    from spectralclustersupertree_amd import synthetic
    from spectralclustersupertree_amd.backend import Device

    dev = Device(0)
    doc = {"what": "matrix-free S X from the flattened tables beside the dense path; one MI355X; times in seconds "
                   "(wall, tables already resident in HBM), the best of --repeat runs", "configs": {}}
    for name in args.configs.split(","):
        n, m, weights = CONFIGS[name]
        tables = synthetic.make_tables(0, n, m, "branch", random_weights=weights)
        dtab = dev.upload(tables)
        entry = {"n_taxa": n, "n_trees": m, "per_tree_weights": weights, "leaves": int(tables.leaf_taxon.size)}
        # ---- dense
        best = None
        for _ in range(args.repeat):
            def dense():
                g = dtab.build()
                maps, st = g.fiedler(None)
                return g, maps, st
            (g, maps, st), dt = timed(dense)
            bst = g.build_stats
            g.free()
            if best is None or dt < best[0]:
                best = (dt, maps, st, bst)
        dt, maps, st, bst = best
        n64 = st["n_apply"] - st["n_apply32"]
        entry["dense"] = {
            "build_plus_solve_s": round(dt, 4), "build_ms": round(bst["total_ms"], 3), "solve_ms": round(st["solve_ms"], 3),
            "block": st["block"], "iterations": st["iterations"], "applications": st["n_apply"],
            "ms_per_application_W": round(st["apply_ms_total"] / max(n64, 1), 4),
            "ms_per_application_image": round(st["apply32_ms_total"] / max(st["n_apply32"], 1), 4) if st["n_apply32"] else None,
            "lambda2": st["lambda"][1], "residual": max(st["resid"]),
        }
        deg = None
        # ---- matrix-free
        for block in (4, 8):
            best = None
            err = None
            for _ in range(args.repeat):
                try:
                    def mf():
                        gm = dtab.matrix_free_graph(max_block=block)
                        maps2, st2 = gm.fiedler(None, block=block)
                        return gm, maps2, st2
                    (gm, maps2, st2), dt2 = timed(mf)
                except nv.ScsError as e:  # e.g. no room for the slabs
                    err = str(e)
                    break
                if deg is None:
                    deg = gm.degrees()
                gm.free()
                if best is None or dt2 < best[0]:
                    best = (dt2, maps2, st2)
            if best is None:
                entry[f"matrix_free_b{block}"] = {"error": err}
                continue
            dt2, maps2, st2 = best
            dd = np.sqrt(deg)
            x, x2 = maps[:, 1] * dd, maps2[:, 1] * dd
            x, x2 = x / np.linalg.norm(x), x2 / np.linalg.norm(x2)
            entry[f"matrix_free_b{block}"] = {
                "setup_plus_solve_s": round(dt2, 4), "solve_ms": round(st2["solve_ms"], 3),
                "iterations": st2["iterations"], "applications": st2["n_apply"],
                "ms_per_application": round(st2["apply_ms_total"] / max(st2["n_apply"], 1), 4),
                "bytes_per_application": st2["apply_bytes"],
                "slabs_GB": round(2.0 * m * n * block * 8 / 2**30, 2),
                "lambda2": st2["lambda"][1], "residual": max(st2["resid"]),
                "distance_to_dense": {
                    "lambda2": abs(st2["lambda"][1] - st["lambda"][1]),
                    "embedding_column_max_abs": float(np.max(np.abs(maps2[:, 1] - maps[:, 1]))),
                    "embedding_column_scale": float(np.max(np.abs(maps[:, 1]))),
                    "unit_norm_eigenvector_max_abs": float(np.max(np.abs(x2 - x))),
                    "labels_by_sign_equal": bool(np.array_equal(maps2[:, 1] > 0, maps[:, 1] > 0)),
                },
                "speedup_over_dense_build_plus_solve": round(dt / dt2, 2),
            }
        dtab.free()
        doc["configs"][name] = entry
        print(json.dumps({name: entry}), flush=True)
    dev.close()
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
