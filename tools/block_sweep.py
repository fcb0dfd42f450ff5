"""LOBPCG block width at a BASELINE.json size: iterations and time of scs_fiedler for b = 4, 8
(and the automatic choice) on the same graph (run on the GPU box).
    python tools/block_sweep.py [taxa] [trees] [seed]"""
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tables = synthetic.make_tables(seed, n, m, "branch")
dev = Device(0)
dt = dev.upload(tables)
g = dt.build()
dt.free()
out = {"taxa": n, "trees": m, "seed": seed, "blocks": {}}
ref = None
for block in (0, 4, 8, 4, 8):
    t0 = time.perf_counter()
    maps, st = g.fiedler(None, block=block)
    dt_s = time.perf_counter() - t0
    if ref is None:
        ref = maps
    out["blocks"].setdefault(str(block), []).append(
        {"s": round(dt_s, 5), "iterations": st["iterations"], "block": st["block"], "resid": max(st["resid"]),
         "max_abs_diff_vs_first": float(np.abs(maps - ref).max())})
print(json.dumps(out, indent=1))
