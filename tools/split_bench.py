"""TreeArrays.split by phase on a deep-recursion-shaped forest (thousands of tiny trees), for a
range of helper thread counts (run on the GPU box: its host cores are what the recursion runs on).
    python tools/split_bench.py [taxa] [trees]"""
import os

os.environ.setdefault("SCS_DEBUG", "1")  # (tools may use the probe switches: csrc/scs_internal.h scs_dbg)
import ctypes as C, os, subprocess, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

if len(sys.argv) > 3:  # child: one thread count
    from spectralclustersupertree_amd import synthetic, _hostlib
    from spectralclustersupertree_amd.treearrays import _p
    n, m = int(sys.argv[1]), int(sys.argv[2])
    lib = _hostlib.load()
    a = synthetic.tree_arrays(1, n, m)
    parts = [np.arange(0, n // 2, dtype=np.int32), np.arange(n // 2, n, dtype=np.int32)]
    part_of = np.full(n, -1, np.int32); new_id = np.zeros(n, np.int32)
    for c, ids in enumerate(parts):
        part_of[ids] = c; new_id[ids] = np.arange(len(ids), dtype=np.int32)
    lc = np.ascontiguousarray(a.leaf_counts(), dtype=np.int64)
    pt = np.zeros(2, np.int64); pn = np.zeros(2, np.int64)
    best = [1e9] * 4
    for rep in range(30):
        plan = C.c_void_p()
        t0 = time.perf_counter()
        lib.scs_host_split_begin(a.n_trees, _p(a.node_off, C.c_int64), _p(a.parent, C.c_int32), _p(a.taxon, C.c_int32),
                                 _p(a.length, C.c_double), _p(a.support, C.c_double), _p(lc, C.c_int64),
                                 _p(part_of, C.c_int32), _p(new_id, C.c_int32), 2, C.byref(plan), _p(pt, C.c_int64),
                                 _p(pn, C.c_int64))
        t1 = time.perf_counter()
        for c in range(2):
            mm, total = int(pt[c]), int(pn[c])
            node_off = np.zeros(mm + 1, np.int64); ti = np.empty(mm, np.int32); lcc = np.zeros(mm, np.int64)
            pres = np.zeros(n, np.uint8)
            par = np.empty(total, np.int32); tax = np.empty(total, np.int32); ln = np.empty(total); su = np.empty(total)
            lib.scs_host_split_fill(plan, c, _p(node_off, C.c_int64), _p(ti, C.c_int32), _p(lcc, C.c_int64),
                                    _p(par, C.c_int32), _p(tax, C.c_int32), _p(ln, C.c_double), _p(su, C.c_double),
                                    _p(pres, C.c_uint8))
        t2 = time.perf_counter()
        lib.scs_host_split_end(plan)
        t3 = time.perf_counter()
        whole0 = time.perf_counter(); a.split(parts); whole = time.perf_counter() - whole0
        for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, whole)):
            best[i] = min(best[i], v)
    print(f"threads {os.environ.get('SCS_HOST_THREADS', 'default'):>7}: begin {best[0]*1e6:7.0f} us  fill x2 {best[1]*1e6:6.0f} us  "
          f"end {best[2]*1e6:4.0f} us | TreeArrays.split {best[3]*1e6:7.0f} us  ({int(a.node_off[-1])} nodes)")
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "16"
    m = sys.argv[2] if len(sys.argv) > 2 else "5000"
    for thr in ("1", "2", "4", "8", "16", "32", "default"):
        env = dict(os.environ)
        if thr != "default":
            env["SCS_HOST_THREADS"] = thr
        else:
            env.pop("SCS_HOST_THREADS", None)
        subprocess.run([sys.executable, __file__, n, m, "child"], env=env, check=False)
