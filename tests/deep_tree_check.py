"""One-off check of the 64-bit key path on naturally deep trees (run on the GPU box):
caterpillar trees with 70 000 leaves have LCA depths up to 69 998, which do not fit next to
a 17-bit position in 32 bits."""
import os as _os

_os.environ.setdefault("SCS_DEBUG", "1")  # (the sweeps force probe paths: csrc/scs_internal.h scs_dbg)
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from oracle import tables_oracle as to
from spectralclustersupertree_amd.backend import Device
from spectralclustersupertree_amd.flatten import TreeTables

n = 70000
rng = np.random.RandomState(0)
offs, taxon, depth, val = [0], [], [], []
for t in range(3):
    perm = rng.permutation(n).astype(np.int32)
    d = np.arange(n, dtype=np.int32)        # LCA(leaf p, leaf p+1) sits at depth p of the comb
    d[-1] = 0                               # padding slot
    taxon.append(perm); depth.append(d); val.append(d.astype(np.float64))
    offs.append(offs[-1] + n)
tables = TreeTables(n_taxa=n, tree_off=np.asarray(offs, dtype=np.int64), leaf_taxon=np.concatenate(taxon),
                    adj_depth=np.concatenate(depth), adj_val=np.concatenate(val),
                    tree_w=np.asarray([1.0, 0.5, 2.0]), taxa=None, monotone=True)
dev = Device(0)
dtab = dev.upload(tables)
t0 = time.perf_counter()
g = dtab.build()
print("build", round(time.perf_counter() - t0, 3), "s", g.build_stats["n_tiles"], "tiles")
rows = np.asarray([0, 1, 777, 35000, 69998, 69999], dtype=np.int32)
want = to.pcg_rows(tables, rows)
bad = 0
for i, r in enumerate(rows):
    got = g.download_rows(int(r), 1)[0]
    bad += int(np.count_nonzero(got != want[i]))
print("cells mismatched:", bad, "max value", want.max())
g.free(); dtab.free(); dev.close()
sys.exit(1 if bad else 0)
