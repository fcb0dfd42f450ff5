"""Generate the golden vectors under tests/golden/vectors/ (run in the build container).

Every vector is produced by the ORACLE side only: the dict-based restatement of the
reference (oracle/scs_oracle.py) for W / contraction, and scikit-learn itself -- the
reference's actual numerics, called exactly as the reference calls it -- for the
embedding and the labels.  Versions are recorded in each file.  The reference package
cannot be imported here (cogent3 missing, PEP 695 syntax), see oracle/scs_oracle.py.

    python tests/golden/make_golden.py
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import scipy  # noqa: E402
import sklearn  # noqa: E402
from reference_cases import DATA_DIR, INLINE_CASES  # noqa: E402

from oracle import scs_oracle as so  # noqa: E402
from oracle import tables_oracle as to  # noqa: E402
from spectralclustersupertree_amd import flatten as fl  # noqa: E402
from spectralclustersupertree_amd import synthetic  # noqa: E402
from spectralclustersupertree_amd.scs import relabel_for_contraction  # noqa: E402
from spectralclustersupertree_amd.tree import make_tree  # noqa: E402

OUT = Path(__file__).resolve().parent / "vectors"
SEED = 12345

ZERO_WEIGHT_CASES = [
    ("zero_weight_branch", ["((a:1,b:1,x:1):0.0,((c:1,d:1):0.5,(e:1,f:1):0.25):1.0)",
                            "((a:1,c:1):1.0,(b:1,(d:1,e:1):0.5):1.0)",
                            "(((a:1,e:1):0.0,f:1):0.0,(b:1,c:1):2.0)",
                            "((x:1,(a:1,b:1):0.0):0.0,(d:1,f:1):1.5)"], "branch"),
    ("zero_weight_bootstrap", ["((a,b,x)0,((c,d)80,(e,f)60)90)", "((a,c)70,(b,(d,e)50)75)",
                               "(((a,e)0,f)0,(b,c)95)", "((x,(a,b)0)0,(d,f)65)"], "bootstrap"),
]


def vector_from_trees(name, trees, weights, strategy, contract):
    names = sorted(so._all_tips(trees))
    vertices = {(n,) for n in names}
    adj, weight, occ, together = so.build_pcg(vertices, trees, weights, strategy)
    w_dict = so.dense_matrix([(n,) for n in names], weight)
    tables = fl.flatten_trees(trees, weights, strategy, names)
    groups = fl.contraction_groups(tables) if contract else np.arange(len(names), dtype=np.int32)
    if contract:
        so.contract_pcg(vertices, adj, weight, occ, together)
    order = sorted(vertices)
    a = so.dense_matrix(order, weight)
    return finish(name, tables, w_dict, groups, a, strategy)


def vector_from_tables(name, tables, strategy):
    w, _ = to.pcg_dense(tables)
    groups = np.arange(tables.n_taxa, dtype=np.int32)
    return finish(name, tables, w, groups, w, strategy)


def finish(name, tables, w, groups, a, strategy):
    maps = to.sign_flip_columns(so.spectral_maps(a, np.random.RandomState(SEED)))
    labels = so.spectral_labels(a, np.random.RandomState(SEED))
    s, dd = to.normalized_operator(a)
    lam = np.sort(np.linalg.eigvalsh(s))[::-1][:3]
    np.savez_compressed(
        OUT / f"{name}.npz",
        n_taxa=tables.n_taxa, tree_off=tables.tree_off, leaf_taxon=tables.leaf_taxon,
        adj_depth=tables.adj_depth, adj_val=tables.adj_val, tree_w=tables.tree_w,
        monotone=tables.monotone, strategy=strategy, w=w, groups=groups, a=a, maps=maps,
        labels=labels, lam=lam, seed=SEED,
        versions=f"scikit-learn {sklearn.__version__}, scipy {scipy.__version__}, numpy {np.__version__}",
    )
    print(f"{name}: V={a.shape[0]} taxa={tables.n_taxa} trees={tables.n_trees} lam={lam}")


def main():
    OUT.mkdir(exist_ok=True)
    by_name = {c.name: c for c in INLINE_CASES}
    for nm in ("simple_inconsistency", "simple_contraction", "depth_depth", "branch_branch",
               "bootstrap_bootstrap", "weights_2_1"):
        c = by_name[nm]
        trees = [make_tree(s) for s in c.trees]
        vector_from_trees(f"inline_{nm}", trees, c.weights or [1.0] * len(trees), c.pcg_weighting,
                          c.contract_edges)
    # round 6 ("the 12 inline cases"): the other inline cases that reach the spectral step
    for nm in ("two_squares", "weights_1001_1", "weights_1_2", "weights_1_1001", "depth_one", "depth_branch",
               "branch_one", "branch_depth", "bootstrap_one", "bootstrap_depth"):
        c = by_name[nm]
        trees = [make_tree(s) for s in c.trees]
        vector_from_trees(f"inline_{nm}", trees, c.weights or [1.0] * len(trees), c.pcg_weighting,
                          c.contract_edges)
    # ... and "one zero-branch-length case": an edge of the proper cluster graph exists wherever two taxa share a
    # root side (scs.py:651-652), whatever its weight (:655-658) -- here taxon x is joined to a and b only through
    # zero-length / zero-support clades: ONE component, and a zero row of W (scikit-learn warns "Graph is not fully
    # connected", sets that vertex's degree factor to 1 and still splits, scipy/sparse/csgraph/_laplacian.py:552-557)
    for nm, newicks, strat in ZERO_WEIGHT_CASES:
        vector_from_trees(nm, [make_tree(s) for s in newicks], [1.0] * len(newicks), strat, True)
    trees = [make_tree(x.strip()) for x in (DATA_DIR / "supertriplets_source.tre").read_text().splitlines() if x.strip()]
    vector_from_trees("fixture_supertriplets_top", trees, [1.0] * len(trees), "depth", True)
    # top-level W of the dcm fixture under `one` and `branch` (two components: the solve never fires there,
    # tests/test_spectral_cluster_supertree.py:50-61, :244-248 -- the vectors pin W and the contraction groups)
    trees = [make_tree(x.strip()) for x in (DATA_DIR / "dcm_source_trees.tre").read_text().splitlines() if x.strip()]
    for strat in ("one", "branch"):
        vector_from_trees(f"fixture_dcm_top_{strat}", trees, [1.0] * len(trees), strat, False)
    vector_from_tables("synthetic_256_branch", synthetic.make_tables(256, 256, 32, "branch", random_weights=True), "branch")
    vector_from_tables("synthetic_1000_depth", synthetic.make_tables(1000, 1000, 100, "depth"), "depth")
    for strat in ("one", "depth", "branch", "bootstrap"):
        vector_from_tables(f"synthetic_64_{strat}", synthetic.make_tables(64, 64, 8, strat, leaves_per_tree=50), strat)
    vector_from_tables("synthetic_200_branch_weighted",
                       synthetic.make_tables(200, 200, 16, "branch", leaves_per_tree=170, random_weights=True), "branch")
    # planted contraction: pairs that always travel together
    trees = [make_tree(s) for s in ["(((a,b),(c,d)),((e,f),(g,h)))", "((a,b),((c,d),(g,(e,f))))",
                                    "(((a,b),e),((c,d),(f,h)))", "((h,(a,b)),((c,d),g))"]]
    vector_from_trees("planted_contraction", trees, [1.0, 2.0, 1.0, 0.5], "branch", True)


if __name__ == "__main__":
    main()
