"""Where the time of a device-side forest split goes (round 5): one parent forest, split again and
again; Python-side wall time of the split call and of the tables download, per call.
    python tools/forest_split_bench.py [taxa] [trees] [reps]      (rocprofv3 --kernel-trace --stats for kernel times)"""
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device
from spectralclustersupertree_amd.treearrays import ResidentArrays, _STRATEGY_CODE

k = int(sys.argv[1]) if len(sys.argv) > 1 else 24
m = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
arr = synthetic.tree_arrays(1, k, m)
dev = Device(0)
res = ResidentArrays.from_host(arr, dev)
parts = [np.arange(0, k // 2, dtype=np.int32), np.arange(k // 2, k, dtype=np.int32)]
part_of = np.full(k, -1, dtype=np.int32); new_id = np.zeros(k, dtype=np.int32)
for c, ids in enumerate(parts):
    part_of[ids] = c; new_id[ids] = np.arange(len(ids), dtype=np.int32)
for _ in range(5):
    res.split(parts, "branch")
t_split = t_tab = t_all = 0.0
for _ in range(reps):
    t0 = time.perf_counter()
    kids = res.forest.split(part_of, new_id, [len(p) for p in parts], _STRATEGY_CODE["branch"])
    t1 = time.perf_counter()
    for f in kids:
        f.tables()
    t2 = time.perf_counter()
    t_split += t1 - t0; t_tab += t2 - t1
    del kids
t0 = time.perf_counter()
for _ in range(reps):
    res.split(parts, "branch")
t_all = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(reps):
    ch = arr.split(parts); [c.flatten("branch") for c in ch]
t_host = time.perf_counter() - t0
print(f"{k} taxa x {m} trees ({len(arr.parent)} nodes): scs_forest_split {t_split / reps * 1e6:.0f} us, tables download "
      f"{t_tab / reps * 1e6:.0f} us (two children), ResidentArrays.split {t_all / reps * 1e6:.0f} us; host split + flatten {t_host / reps * 1e6:.0f} us")
