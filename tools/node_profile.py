"""Where one recursion node's time goes, by problem size (run on the GPU box)."""
import sys, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from sklearn.cluster import k_means
from spectralclustersupertree_amd import scs, synthetic, flatten as fl
from spectralclustersupertree_amd.treearrays import TreeArrays
warnings.simplefilter("ignore")
dev = scs.default_device()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100
arr = synthetic.tree_arrays(1, 5000, M, 4000)
from spectralclustersupertree_amd import kmeans2
rng = np.random.RandomState(0)
def T(f, n=10):
    f()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
for size in (8, 20, 60, 100, 200, 1000):
    keep = np.sort(rng.choice(5000, size, replace=False))
    sub = arr.restrict(keep)
    present = sub.present_taxa()
    tab = sub.flatten("branch", local_ids=present)
    n = tab.n_taxa
    t_groups, _ = T(lambda: (fl.pcg_components(tab), fl.contraction_groups(tab)))
    t_up, dtab = T(lambda: dev.upload(tab))
    t_build, g = T(lambda: dtab.build())
    v0 = rng.uniform(-1, 1, n)
    t_fied, (maps, st) = T(lambda: g.fiedler(v0))
    t_km, _ = T(lambda: k_means(maps, 2, random_state=np.random.RandomState(0), n_init=10))
    import threadpoolctl
    with threadpoolctl.threadpool_limits(1):
        t_km1, _ = T(lambda: k_means(maps, 2, random_state=np.random.RandomState(0), n_init=10))
    t_all, _ = T(lambda: scs.spectral_bipartition_device(tab, np.random.RandomState(0), contract_edges=True))
    t_km2, _ = T(lambda: kmeans2.labels(maps, np.random.RandomState(0)))
    print(f"V={n:5d} trees={sub.n_trees:3d} kmeans2 {t_km2:.2f} groups {t_groups:.2f} upload {t_up:.2f} build {t_build:.2f} fiedler {t_fied:.2f} "
          f"kmeans {t_km:.2f} (1 thread {t_km1:.2f}) | whole node {t_all:.2f} ms")
