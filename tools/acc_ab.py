#!/usr/bin/env python3
"""A/B of the two monotone tile kernels on one device: the 4-wave kernel (one 64 x 256 tile per
workgroup, scs_mono.h; SCS_WIDE=0) against the producer / consumer kernel (two tiles of a row block
per workgroup, scs_mono_wide.h; SCS_WIDE=1), same tables, same process.  Prints accumulate ms, cell-trees/s and -- last
repetition -- sampled rows against the C oracle and the two matrices against each other.

    python tools/acc_ab.py [n_taxa n_trees [reps]]      (default 10000 500 5)
    SCS_ACC_STAMP=1 python tools/acc_ab.py ...           phase stamps of either kernel on stderr
"""
import json
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from oracle import tables_oracle as to  # noqa: E402
from spectralclustersupertree_amd import synthetic  # noqa: E402
from spectralclustersupertree_amd.backend import Device  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    modes = os.environ.get("ACC_AB_MODES", "0,1,0,1").split(",")
    tables = synthetic.make_tables(0, n, m, "branch", pinned=True)
    rows = np.unique(np.random.RandomState(1).randint(0, n, size=12)).astype(np.int32)
    want = to.pcg_rows(tables, rows)
    out = {"n": n, "m": m, "runs": []}
    with Device(0) as dev:
        dtab = dev.upload(tables)
        dev.synchronize()
        for wide in modes:
            os.environ["SCS_WIDE"] = wide
            acc, prep = [], []
            bad = None
            for r in range(reps):
                g = dtab.build()
                acc.append(g.build_stats["accumulate_ms"])
                prep.append(g.build_stats["prep_ms"])
                if r == reps - 1:
                    bad = sum(int(np.count_nonzero(g.download_rows(int(x), 1)[0] != want[i]))
                              for i, x in enumerate(rows))
                    cells = g.build_stats["cell_trees"]
                g.free()
            best = min(acc[1:] or acc)
            out["runs"].append({"wide": wide, "accumulate_ms": [round(a, 3) for a in acc],
                                "prep_ms": round(float(np.median(prep)), 3),
                                "cell_trees_per_s": round(cells / (best * 1e-3), 0),
                                "cells_mismatched_vs_oracle": bad})
        dtab.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
