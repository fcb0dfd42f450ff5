"""The table-level oracle (flatten + oracle/pcg_oracle.c + host graph helpers)
against the dict-based restatement of the reference, bit for bit, on the
reference's own fixtures and inline cases and on synthetic ragged inputs."""

import numpy as np
import pytest
from reference_cases import DATA_DIR, FILE_CASES, INLINE_CASES

from oracle import scs_oracle as so
from oracle import tables_oracle as to
from spectralclustersupertree_amd import flatten as fl
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.scs import relabel_for_contraction
from spectralclustersupertree_amd.tree import make_tree


def _dict_side(trees, weights, strategy):
    names = sorted(so._all_tips(trees))
    vertices = {(n,) for n in names}
    adj, weight, occ, together = so.build_pcg(vertices, trees, weights, strategy)
    order = [(n,) for n in names]
    return names, vertices, adj, weight, occ, together, so.dense_matrix(order, weight)


def _check(trees, weights, strategy):
    names, vertices, adj, weight, occ, together, dense = _dict_side(trees, weights, strategy)
    tables = fl.flatten_trees(trees, weights, strategy, names)
    tables.validate()
    w, updates = to.pcg_dense(tables)
    assert updates == sum(together.values())
    assert np.array_equal(w, dense), "dense W differs from the dict-based restatement"
    # occurrences
    assert np.array_equal(fl.taxa_occurrences(tables), [occ[(n,)] for n in names])
    # components: same partition
    comps = so.graph_components(set(vertices), adj)
    lab = fl.pcg_components(tables)
    got = {}
    for i, c in enumerate(lab):
        got.setdefault(int(c), set()).add((names[i],))
    assert sorted(map(sorted, got.values())) == sorted(map(sorted, comps))
    # contraction: same groups, same contracted matrix
    if len(comps) == 1:
        so.contract_pcg(vertices, adj, weight, occ, together)
        order = sorted(vertices)
        groups = fl.contraction_groups(tables)
        n_groups = int(groups.max()) + 1
        assert n_groups == len(order)
        members = [tuple(sorted(names[i] for i in np.flatnonzero(groups == g))) for g in range(n_groups)]
        assert members == order
        work, perm, group_start = relabel_for_contraction(tables, groups)
        w2, _ = to.pcg_dense(work)
        assert np.array_equal(w2, w[np.ix_(perm, perm)])
        contracted = to.contract_dense(w2, group_start)
        assert np.array_equal(contracted, so.dense_matrix(order, weight))


@pytest.mark.parametrize("case", INLINE_CASES, ids=lambda c: c.name)
def test_inline_tables(case):
    trees = [make_tree(s) for s in case.trees]
    if len(so._all_tips(trees)) < 2:
        pytest.skip("trivial")
    weights = case.weights or [1.0] * len(trees)
    _check(trees, weights, case.pcg_weighting)


@pytest.mark.parametrize(("name", "src", "exp", "weighting"), FILE_CASES, ids=[c[0] for c in FILE_CASES])
def test_fixture_tables(name, src, exp, weighting):
    trees = [make_tree(x.strip()) for x in (DATA_DIR / src).read_text().splitlines() if x.strip()]
    _check(trees, [1.0] * len(trees), weighting)


@pytest.mark.parametrize("strategy", ["one", "depth", "branch", "bootstrap"])
def test_synthetic_ragged_tables(strategy):
    trees = synthetic.tree_objects(11, 60, 9, leaves_per_tree=37)
    weights = [1.0, 2.0, 0.5, 1.25, 1.0, 3.0, 1.0, 0.75, 1.5]
    _check(trees, weights, strategy)


def test_contraction_merges_planted():
    # a and b always together; c, d always together in the trees holding them
    trees = [make_tree(s) for s in ["(((a,b),(c,d)),(e,f))", "((a,b),((c,d),g))", "(((a,b),e),(c,d))"]]
    _check(trees, [1.0, 1.0, 1.0], "depth")


@pytest.mark.parametrize("strategy", ["depth", "bootstrap"])
def test_all_core_oracle_build_gives_the_same_bits(strategy):
    # bench.py's all-core cpu_baseline leg: rows partitioned over threads, tree order per cell
    from spectralclustersupertree_amd import synthetic

    tables = synthetic.make_tables(9, 257, 12, strategy, leaves_per_tree=200, random_weights=True)
    w1, _ = to.pcg_dense(tables)
    for threads in (1, 3, 8):
        assert np.array_equal(to.pcg_dense_mt(tables, threads), w1)


def test_planted_sets_are_model_tree_plus_moves():
    # SURVEY.md 8d planted variant: same model tree behind every source tree of a set
    from spectralclustersupertree_amd import synthetic

    a = synthetic.make_tables(5, 200, 6, "depth", planted_spr=0)
    # no moves: every tree IS the model tree (same leaf order, same LCA depths)
    k = 200
    for t in range(1, 6):
        assert np.array_equal(a.leaf_taxon[:k], a.leaf_taxon[t * k:(t + 1) * k])
        assert np.array_equal(a.adj_depth[:k], a.adj_depth[t * k:(t + 1) * k])
    b = synthetic.make_tables(5, 200, 6, "depth", planted_spr=4)
    b.validate()
    assert sorted(b.leaf_taxon[:k].tolist()) == list(range(k))
    assert not np.array_equal(b.adj_depth[:k], b.adj_depth[k:2 * k])
    assert int(fl.pcg_components(b).max()) == 0
