import sys, time, numpy as np
sys.path.insert(0,'.')
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device
n,m=50000,2000
tables = synthetic.make_tables(0, n, m, "branch")
v0 = np.random.RandomState(0).uniform(-1, 1, n)
with Device(0) as dev:
    dtab = dev.upload(tables); dev.synchronize()
    for rep in range(3):
        t0=time.perf_counter(); g = dtab.build(); dev.synchronize(); t1=time.perf_counter()
        maps, st = g.fiedler(v0); dev.synchronize(); t2=time.perf_counter()
        g.free(); dev.synchronize(); t3=time.perf_counter()
        print(f"build wall {t1-t0:.3f} (dev {g.build_stats['total_ms']:.0f} ms) fiedler wall {t2-t1:.3f} (dev {st['solve_ms']:.0f} ms) free {t3-t2:.3f}")
    dtab.free()
