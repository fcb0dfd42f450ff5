"""Host wall time of every call of one protocol step against the device time the library reports
(run on the GPU box): where the step spends what the kernels do not.
    python tools/overhead_check.py [taxa] [trees]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
tables = synthetic.make_tables(0, n, m, "branch", pinned=True)
v0 = np.random.RandomState(0).uniform(-1, 1, n)
with Device(0) as dev:
    for rep in range(6):
        dev.synchronize()
        t0 = time.perf_counter(); dtab = dev.upload(tables)
        t1 = time.perf_counter(); g = dtab.build()
        t2 = time.perf_counter(); maps, st = g.fiedler(v0)
        t3 = time.perf_counter(); g.free()
        t4 = time.perf_counter(); dtab.free()
        t5 = time.perf_counter()
        b = g.build_stats
        print(f"step {1e3*(t5-t0):7.3f} ms | upload {1e3*(t1-t0):6.3f} | build {1e3*(t2-t1):6.3f} (device {b['total_ms']:.3f}: prep "
              f"{b['prep_ms']:.3f} acc {b['accumulate_ms']:.3f}) | fiedler {1e3*(t3-t2):6.3f} (loop {st['solve_ms']:.3f}, "
              f"{st['iterations']} it) | graph.free {1e3*(t4-t3):6.3f} | tables.free {1e3*(t5-t4):6.3f}")
