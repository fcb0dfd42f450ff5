"""One mid-size recursion node by stage (run on the GPU box): python tools/node_profile2.py [trees]"""
import sys, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import scs, synthetic, flatten as fl, kmeans2
warnings.simplefilter("ignore")
dev = scs.default_device()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
def T(f, n=8):
    f()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
for size in (80, 120, 200, 400, 1000, 3000):
    tab = synthetic.make_tables(3, size, M, "branch")
    t_groups, _ = T(lambda: (fl.pcg_components(tab), fl.contraction_groups(tab)))
    t_up, dtab = T(lambda: dev.upload(tab))
    t_build, g = T(lambda: dtab.build())
    v0 = np.random.RandomState(0).uniform(-1, 1, size)
    t_fied, (maps, st) = T(lambda: g.fiedler(v0))
    t_km2, _ = T(lambda: kmeans2.labels(maps, np.random.RandomState(0)))
    t_all, _ = T(lambda: scs.spectral_bipartition_device(tab, np.random.RandomState(0), contract_edges=True))
    print(f"V={size:5d} trees={M} groups {t_groups:.2f} upload {t_up:.2f} build {t_build:.2f} (dev {g.build_stats['total_ms']:.2f}) "
          f"fiedler {t_fied:.2f} (dev {st['solve_ms']:.2f}, {st['iterations']} it, b={st['block']}) kmeans2 {t_km2:.2f} | whole node {t_all:.2f} ms", flush=True)
