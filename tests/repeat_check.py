"""W must be the same bits on every build (and the oracle's): builds interleaved with solves."""
import os as _os

_os.environ.setdefault("SCS_DEBUG", "1")  # (the sweeps force probe paths: csrc/scs_internal.h scs_dbg)
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import tables_oracle as to
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

for (n, m, st) in [(3000, 120, "branch"), (10000, 500, "branch")]:
    tables = synthetic.make_tables(0, n, m, st)
    v0 = np.random.RandomState(0).uniform(-1, 1, n)
    with Device(0) as dev:
        dtab = dev.upload(tables)
        ws, ms = [], []
        for rep in range(4):
            g = dtab.build()
            ws.append(g.download())
            maps, stats = g.fiedler(v0)
            ms.append((maps, stats["lambda"][1], stats["iterations"]))
            g.free()
        dtab.free()
    print(n, "W repeat diffs", [int(np.count_nonzero(ws[0] != w)) for w in ws[1:]], "sym",
          bool(np.array_equal(ws[0], ws[0].T)))
    print("   lambda2", [x[1] for x in ms], "iters", [x[2] for x in ms], "maps equal",
          [bool(np.array_equal(ms[0][0], x[0])) for x in ms[1:]])
    for w in ws[1:]:
        d = np.argwhere(ws[0] != w)
        if len(d):
            print("   first diffs", d[:6].tolist(), ws[0][tuple(d[0])], w[tuple(d[0])])
    if n <= 3000:
        ref, _ = to.pcg_dense(tables)
        print("   vs oracle mismatches", int(np.count_nonzero(ws[0] != ref)))
    else:
        rows = np.arange(0, n, 97, dtype=np.int32)
        want = to.pcg_rows(tables, rows)
        print("   rows vs oracle mismatches", int(np.count_nonzero(ws[0][rows] != want)), "of", want.size)
