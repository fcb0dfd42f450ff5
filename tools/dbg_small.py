import sys, numpy as np
sys.path.insert(0,'.')
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device
from oracle import tables_oracle as to
rs = np.random.RandomState(4)
nodes = []
for i in range(12):
    n = int(rs.randint(3, 65))
    nodes.append((synthetic.make_tables(500 + i, n, int(rs.randint(2, 30)), ["one", "depth", "branch", "bootstrap"][i % 4],
                                        leaves_per_tree=max(2, n - int(rs.randint(0, 3))),
                                        random_weights=bool(i % 2)), None))
with Device(0) as dev:
    out = dev.small_solve(nodes, want_w=True)
    for i,((tables, _), (maps, lam, w)) in enumerate(zip(nodes, out)):
        ref,_ = to.pcg_dense(tables)
        dtab = dev.upload(tables); g = dtab.build(); wb = g.download(); g.free(); dtab.free()
        d1 = np.argwhere(w != ref); d2 = np.argwhere(wb != ref)
        print(i, tables.n_taxa, tables.n_trees, "fused!=oracle", len(d1), "big!=oracle", len(d2))
        if len(d1): 
            a,b = d1[0]; print("   ", a,b, w[a,b], ref[a,b], (w[a,b]-ref[a,b])/ref[a,b])
