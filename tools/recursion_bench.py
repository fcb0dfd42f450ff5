"""Whole-recursion timing of construct_supertree on a synthetic input (run on the GPU box).

Compares the two host paths of the recursion -- flat tree arrays (libscs_host.so, the
product path) and Python tree objects (the reference's way, kept as _construct_objects) --
with the same device bipartition; prints one JSON line.

    python tools/recursion_bench.py [--taxa 2000] [--trees 50] [--leaves 1500] [--strategy branch]
"""
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import argparse, json, sys, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import scs, synthetic
from spectralclustersupertree_amd.treearrays import TreeArrays

ap = argparse.ArgumentParser()
ap.add_argument("--taxa", type=int, default=2000)
ap.add_argument("--trees", type=int, default=50)
ap.add_argument("--leaves", type=int, default=None)
ap.add_argument("--strategy", default="branch")
ap.add_argument("--skip-objects", action="store_true")
ap.add_argument("--native-arrays", action="store_true",
                help="build the flat tree arrays in C (synthetic.tree_arrays): no tree objects at all; implies --skip-objects")
args = ap.parse_args()

if args.native_arrays:
    args.skip_objects = True
    trees = None
else:
    trees = synthetic.tree_objects(1, args.taxa, args.trees, args.leaves)
    weights = [1.0] * len(trees)
device_s = [0.0]
calls = [0]
sizes = []
real = scs.spectral_bipartition_device
real_presolve = scs._presolve_small_children
presolve_s = [0.0]

buckets = {}  # size class -> [nodes, seconds inside spectral_bipartition_device]


def timed(tables, rs, *, contract_edges, **kw):
    t0 = time.perf_counter()
    out = real(tables, rs, contract_edges=contract_edges, **kw)
    dt = time.perf_counter() - t0
    device_s[0] += dt
    calls[0] += 1
    sizes.append(tables.n_taxa)
    n = tables.n_taxa
    key = "<=64" if n <= 64 else "65-512" if n <= 512 else "513-4096" if n <= 4096 else ">4096"
    b = buckets.setdefault(key, [0, 0.0])
    b[0] += 1
    b[1] += dt
    return out

def timed_presolve(*a, **kw):
    t0 = time.perf_counter()
    real_presolve(*a, **kw)
    presolve_s[0] += time.perf_counter() - t0

# the product path: no hook handed down (the recursion then batches the small siblings)
scs.spectral_bipartition_device = timed
scs._presolve_small_children = timed_presolve

warnings.simplefilter("ignore")
scs.default_device()  # context creation outside the timings
res = {"taxa": args.taxa, "trees": args.trees, "leaves_per_tree": args.leaves or args.taxa, "strategy": args.strategy}
t0 = time.perf_counter()
if args.native_arrays:
    arrays = synthetic.tree_arrays(1, args.taxa, args.trees, args.leaves)
else:
    names = sorted(scs._all_tip_names(trees))
    arrays = TreeArrays.from_trees(trees, weights, names)
t_conv = time.perf_counter() - t0
t0 = time.perf_counter()
got = scs._construct(arrays, args.strategy, True, np.random.RandomState(0))
t_arr = time.perf_counter() - t0
res["arrays"] = {"total_s": round(t_arr + t_conv, 3), "convert_once_s": round(t_conv, 3),
                 "device_calls": calls[0], "in_bipartition_s": round(device_s[0], 3),
                 "presolve_batches_s": round(presolve_s[0], 3),
                 "per_node_ms": round(1e3 * (device_s[0] + presolve_s[0]) / max(calls[0], 1), 3),
                 "host_recursion_s": round(t_arr - device_s[0] - presolve_s[0], 3),
                 "small_path": scs._small_path(), "largest_problems": sorted(sizes, reverse=True)[:5],
                 "bipartition_by_taxa": {k: {"nodes": v[0], "s": round(v[1], 3)} for k, v in sorted(buckets.items())}}
if not args.skip_objects:
    device_s[0], calls[0] = 0.0, 0
    scs.spectral_bipartition_device = real
    t0 = time.perf_counter()
    want = scs._construct_objects(trees, weights, args.strategy, True, np.random.RandomState(0), timed)
    t_obj = time.perf_counter() - t0
    res["objects"] = {"total_s": round(t_obj, 3), "device_calls": calls[0], "in_bipartition_s": round(device_s[0], 3),
                      "host_recursion_s": round(t_obj - device_s[0], 3)}
    res["same_supertree"] = bool(got.sorted().same_shape(want.sorted()))
print(json.dumps(res))
