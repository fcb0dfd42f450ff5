"""Where the wall time of a whole recursion goes, by phase, WITHOUT a profiler (perf_counter
around the handful of calls a node makes; run on the GPU box).
    python tools/node_phases.py [taxa] [trees]"""
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import scs, synthetic, kmeans2, flatten as fl
from spectralclustersupertree_amd.treearrays import ResidentArrays, TreeArrays
from spectralclustersupertree_amd.backend import Device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
acc = {}


def timed(name, fn):
    def wrapper(*a, **kw):
        t = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            e = acc.setdefault(name, [0, 0.0])
            e[0] += 1
            e[1] += time.perf_counter() - t
    return wrapper


TreeArrays.split = timed("split", TreeArrays.split)
TreeArrays.flatten = timed("flatten", TreeArrays.flatten)
TreeArrays.present_taxa = timed("present_taxa", TreeArrays.present_taxa)
ResidentArrays.split = timed("split(device)", ResidentArrays.split)
ResidentArrays.flatten = timed("flatten(device child)", ResidentArrays.flatten)
ResidentArrays.from_host = timed("forest upload", ResidentArrays.from_host)
fl.pcg_components = timed("pcg_components", fl.pcg_components)
scs.prepare_node = timed("prepare_node", scs.prepare_node)
Device.small_solve = timed("small_solve", Device.small_solve)
scs._solve_node = timed("solve_node(walk thread)", scs._solve_node)
kmeans2.labels = timed("kmeans", kmeans2.labels)
scs._presolve_small_children = timed("presolve(total)", scs._presolve_small_children)
scs.spectral_bipartition_device = timed("bipartition(total)", scs.spectral_bipartition_device)
scs.connect_trees = timed("connect_trees", scs.connect_trees)
scs.tip_names_to_tree = timed("tip_names_to_tree", scs.tip_names_to_tree)

arrays = synthetic.tree_arrays(1, n, m)
scs.default_device()
scs._construct(synthetic.tree_arrays(2, 300, 20), "branch", True, np.random.RandomState(0))
acc.clear()
t0 = time.perf_counter()
scs._construct(arrays, "branch", True, np.random.RandomState(0))
total = time.perf_counter() - t0
out = {"taxa": n, "trees": m, "total_s": round(total, 3), "ahead": scs._last_ahead_stats,
       "phases": {k: {"calls": v[0], "s": round(v[1], 3), "us_per_call": round(v[1] / max(v[0], 1) * 1e6, 1)}
                  for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])}}
print(json.dumps(out, indent=1))
