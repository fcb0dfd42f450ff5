"""Latency of one batched small-node solve by node size: wall time of begin + end per call.
    python tools/small_solve_bench.py [trees] [reps] [sizes ...]        (rocprofv3 --kernel-trace + tools/trace_tail.py for the launches)"""
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import scs, synthetic
from spectralclustersupertree_amd.backend import Device

m = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = Device(0)
sizes = [int(x) for x in sys.argv[3:]] or [8, 16, 32, 64, 100, 128]
for k in sizes:
    tables = synthetic.make_tables(3, k, m, "branch", random_weights=True)
    work, perm, group_start, n_groups = scs.prepare_node(tables, True)
    for _ in range(5):
        dev.small_solve([(work, group_start)], want_w=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.small_solve([(work, group_start)], want_w=False)
    dt = (time.perf_counter() - t0) / reps
    print(f"{k:4d} taxa x {m} trees ({n_groups} vertices): {dt * 1e6:7.0f} us a solve", flush=True)
dev.close()
