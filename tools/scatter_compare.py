#!/usr/bin/env python3
"""The atomic-scatter comparison variant beside the ordered tile kernel (run on the GPU box).

    python tools/scatter_compare.py [--taxa 10000] [--trees 500] [--strategy branch] [--repeat 2]

One JSON line: build times of both formulations on the same tables, the largest relative
difference of the two matrices, and the pair updates the scatter performed (each one two
8-byte read-modify-writes at random addresses of W: SURVEY.md section 8d's 16 U bytes)."""
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import argparse, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

ap = argparse.ArgumentParser()
ap.add_argument("--taxa", type=int, default=10000)
ap.add_argument("--trees", type=int, default=500)
ap.add_argument("--strategy", default="branch")
ap.add_argument("--repeat", type=int, default=2)
args = ap.parse_args()
tables = synthetic.make_tables(0, args.taxa, args.trees, args.strategy)
with Device(0) as dev:
    dtab = dev.upload(tables)
    res = {"taxa": args.taxa, "trees": args.trees, "strategy": args.strategy}
    for name, kw in (("ordered_tile_kernel", {}), ("atomic_scatter", {"scatter": True})):
        best = None
        for _ in range(args.repeat):
            g = dtab.build(**kw)
            st = g.build_stats
            if best is None or st["total_ms"] < best["total_ms"]:
                best = st
            if _ + 1 < args.repeat:
                g.free()
        res[name] = {"build_ms": round(best["total_ms"], 3), "accumulate_ms": round(best["accumulate_ms"], 3)}
        if name == "ordered_tile_kernel":
            w = g.download()
        else:
            ws = g.download()
        g.free()
    dtab.free()
nz = w != 0
res["max_relative_difference"] = float(np.max(np.abs(ws[nz] - w[nz]) / np.abs(w[nz])))
res["same_zero_pattern"] = bool(np.array_equal(ws != 0, nz))
res["scatter_symmetric"] = bool(np.array_equal(ws, ws.T))
updates = float(np.count_nonzero(np.triu(nz, 1))) if args.trees == 1 else None
res["pair_updates_estimate"] = 0.336 * args.taxa * args.taxa * args.trees
res["rmw_bytes_estimate"] = 16.0 * res["pair_updates_estimate"] * 2  # both triangles
res["slowdown"] = round(res["atomic_scatter"]["accumulate_ms"] / res["ordered_tile_kernel"]["accumulate_ms"], 1)
print(json.dumps(res))
