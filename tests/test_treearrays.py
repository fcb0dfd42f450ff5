"""Flat tree arrays (libscs_host.so) against the tree-object path, bit for bit (CPU).

The array recursion replaces ``get_sub_tree`` + ``flatten_trees`` on Python objects
(reference: src/sc_supertree/scs.py:411-455 and :495-663).  Everything here compares the two
on the same inputs: the induced forests' device tables, repeated restrictions, and the whole
recursion driven by the same (CPU, oracle-based) bipartition.
"""

from __future__ import annotations

import os
import random
from pathlib import Path

import numpy as np
import pytest

from oracle import scs_oracle as so
from oracle import tables_oracle as to
from spectralclustersupertree_amd import flatten as fl
from spectralclustersupertree_amd import scs
from spectralclustersupertree_amd.tree import TreeNode, make_tree
from spectralclustersupertree_amd.treearrays import TreeArrays
from tests.reference_cases import INLINE_CASES


def random_tree(rng: random.Random, names: list[str], *, multifurcate=0.3, none_len=0.2, none_sup=0.2,
                unary=0.1, neg_len=0.0) -> TreeNode:
    """Random rooted tree over ``names``: multifurcations, unary nodes, missing lengths/supports."""
    nodes = [TreeNode(n, None, rng.choice([None, round(rng.expovariate(10.0), 6)])
                      if rng.random() < 0.5 else rng.expovariate(10.0)) for n in names]
    rng.shuffle(nodes)
    while len(nodes) > 1:
        k = 2 if rng.random() > multifurcate else rng.randint(3, 4)
        k = min(k, len(nodes))
        kids = [nodes.pop(rng.randrange(len(nodes))) for _ in range(k)]
        ln = None if rng.random() < none_len else rng.expovariate(10.0)
        if ln is not None and rng.random() < neg_len:
            ln = -ln
        sp = None if rng.random() < none_sup else float(rng.randint(50, 100))
        node = TreeNode("", kids, ln, sp)
        if rng.random() < unary:
            node = TreeNode("", [node], rng.expovariate(10.0), sp)
        nodes.append(node)
    root = nodes[0]
    root.length = None
    return root


def random_forest(seed: int, n_taxa: int, n_trees: int, **kw):
    rng = random.Random(seed)
    taxa = [f"t{i:04d}" for i in range(n_taxa)]
    trees, weights = [], []
    for _ in range(n_trees):
        k = rng.randint(max(2, n_taxa // 3), n_taxa)
        trees.append(random_tree(rng, rng.sample(taxa, k), **kw))
        weights.append(rng.choice([1.0, 2.0, 0.5, rng.random() + 0.1]))
    return taxa, trees, weights


def tables_equal(a: fl.TreeTables, b: fl.TreeTables) -> None:
    assert a.n_taxa == b.n_taxa
    assert np.array_equal(a.tree_off, b.tree_off)
    assert np.array_equal(a.leaf_taxon, b.leaf_taxon)
    assert np.array_equal(a.adj_depth, b.adj_depth)
    # bit for bit, signed zeros included
    assert np.array_equal(a.adj_val.view(np.uint64), b.adj_val.view(np.uint64))
    assert np.array_equal(a.tree_w, b.tree_w)
    assert a.monotone == b.monotone


def object_tables(trees, weights, strategy):
    names = sorted(scs._all_tip_names(trees))
    return fl.flatten_trees(trees, weights, strategy, names)


@pytest.mark.parametrize("strategy", ["one", "depth", "branch"])
@pytest.mark.parametrize("seed", range(6))
def test_flatten_matches_object_path(seed, strategy):
    taxa, trees, weights = random_forest(seed, 40, 7, neg_len=0.1 if seed % 2 else 0.0)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    want = fl.flatten_trees(trees, weights, strategy, taxa)
    tables_equal(arrays.flatten(strategy), want)


def test_flatten_bootstrap_and_missing_support():
    taxa, trees, weights = random_forest(3, 25, 5, none_sup=0.0, unary=0.0)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    tables_equal(arrays.flatten("bootstrap"), fl.flatten_trees(trees, weights, "bootstrap", taxa))
    trees[2].children[0].support = None if trees[2].children[0].children else trees[2].children[0].support
    target = next(n for n in trees[2].iter_nontips() if len(n.children) >= 2)
    target.support = None
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    with pytest.raises(TypeError, match="NoneType"):
        fl.flatten_trees(trees, weights, "bootstrap", taxa)
    with pytest.raises(TypeError, match="NoneType"):
        arrays.flatten("bootstrap")


@pytest.mark.parametrize("strategy", ["one", "depth", "branch", "bootstrap"])
@pytest.mark.parametrize("seed", range(8))
def test_restriction_matches_get_sub_tree(seed, strategy):
    rng = random.Random(1000 + seed)
    kw = {"none_sup": 0.0} if strategy == "bootstrap" else {}
    taxa, trees, weights = random_forest(seed, 60, 9, **kw)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    for frac in (0.8, 0.5, 0.2, 0.06):
        keep = sorted(rng.sample(range(len(taxa)), max(2, int(frac * len(taxa)))))
        names = {taxa[i] for i in keep}
        sub_trees, sub_weights = scs._induce(names, trees, weights)
        sub = arrays.restrict(np.asarray(keep))
        assert sub.n_trees == len(sub_trees)
        assert np.array_equal(sub.weights, np.asarray(sub_weights, dtype=np.float64))
        if not sub_trees:
            continue
        present = sub.present_taxa()
        assert [taxa[int(i)] for i in present] == sorted(scs._all_tip_names(sub_trees))
        tables_equal(sub.flatten(strategy, local_ids=present), object_tables(sub_trees, sub_weights, strategy))
        # the induced forest itself (structure, merged lengths, supports) is the object path's
        again = TreeArrays.from_trees(sub_trees, sub_weights, taxa)
        assert np.array_equal(sub.node_off, again.node_off)
        assert np.array_equal(sub.parent, again.parent)
        assert np.array_equal(sub.taxon, again.taxon)
        assert np.array_equal(sub.support.view(np.uint64), again.support.view(np.uint64))
        assert np.array_equal(sub.length.view(np.uint64), again.length.view(np.uint64))


def test_repeated_restriction_equals_single():
    rng = random.Random(7)
    taxa, trees, weights = random_forest(11, 80, 10)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    keep1 = sorted(rng.sample(range(80), 50))
    keep2 = sorted(rng.sample(keep1, 20))
    twice = arrays.restrict(np.asarray(keep1)).restrict(np.asarray(keep2))
    names1, names2 = {taxa[i] for i in keep1}, {taxa[i] for i in keep2}
    t1, w1 = scs._induce(names1, trees, weights)
    t2, w2 = scs._induce(names2, t1, w1)
    present = twice.present_taxa()
    tables_equal(twice.flatten("branch", local_ids=present), object_tables(t2, w2, "branch"))


def test_ragged_edge_cases():
    taxa = ["a", "b", "c", "d"]
    trees = [make_tree("((a,b),(c,d));"), make_tree("(a,b,c);"), make_tree("(d);")]
    arrays = TreeArrays.from_trees(trees, [1.0, 2.0, 3.0], taxa)
    assert arrays.leaf_counts().tolist() == [4, 3, 1]
    # a tree left with fewer than two leaves is dropped (reference: scs.py:447-448)
    sub = arrays.restrict(np.asarray([2, 3]))
    assert sub.n_trees == 1 and sub.weights.tolist() == [1.0]
    assert sub.to_tree(0).sorted().same_shape(make_tree("(c,d);").sorted())
    # nothing left at all
    empty = arrays.restrict(np.asarray([3]))
    assert empty.n_trees == 0 and len(empty.present_taxa()) == 0
    with pytest.raises(ValueError, match="Invalid weighting strategy"):
        arrays.flatten("nope")


def cpu_bipartition(tables, random_state, *, contract_edges):
    """The device step's contract, computed with the oracle (scikit-learn) on the CPU."""
    n = tables.n_taxa
    groups = fl.contraction_groups(tables) if contract_edges else np.arange(n, dtype=np.int32)
    n_groups = int(groups.max()) + 1
    if n_groups < n:
        work, perm, group_start = scs.relabel_for_contraction(tables, groups)
    else:
        work, perm, group_start = tables, np.arange(n, dtype=np.int32), None
    w, _ = to.pcg_dense(work)
    if group_start is not None:
        w = to.contract_dense(w, group_start)
    labels = so.spectral_labels(w, random_state)
    if group_start is None:
        members = [np.array([i], dtype=np.int32) for i in range(n)]
    else:
        members = [perm[group_start[g]: group_start[g + 1]] for g in range(n_groups)]
    return members, np.asarray(labels)


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("strategy", ["one", "branch"])
def test_recursion_on_arrays_equals_recursion_on_objects(seed, strategy):
    import warnings

    taxa, trees, weights = random_forest(50 + seed, 24, 6, unary=0.0)
    names = sorted(scs._all_tip_names(trees))
    arrays = TreeArrays.from_trees(trees, weights, names)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = scs._construct(arrays, strategy, True, np.random.RandomState(seed), cpu_bipartition)
        want = scs._construct_objects(trees, weights, strategy, True, np.random.RandomState(seed),
                                      cpu_bipartition)
    assert got.sorted().same_shape(want.sorted()), f"{got.get_newick()} != {want.get_newick()}"
    assert sorted(got.get_tip_names()) == names


@pytest.mark.parametrize("case", INLINE_CASES, ids=lambda c: c.name)
def test_reference_cases_through_the_array_recursion(case):
    """The reference's own known-answer cases, array recursion + oracle bipartition (CPU)."""
    import warnings

    trees = [make_tree(s) for s in case.trees]
    weights = case.weights if case.weights is not None else [1.0] * len(trees)
    names = sorted(scs._all_tip_names(trees))
    arrays = TreeArrays.from_trees(trees, weights, names)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = scs._construct(arrays, case.pcg_weighting, case.contract_edges, np.random.RandomState(0),
                             cpu_bipartition)
    assert got.sorted().same_shape(make_tree(case.expected).sorted())


# ---------------------------------------------------------------------------
# Newick file -> arrays in C (reference: load.py:7-23)
# ---------------------------------------------------------------------------
def arrays_equal(a: TreeArrays, b: TreeArrays) -> None:
    assert a.taxa == b.taxa
    assert np.array_equal(a.node_off, b.node_off)
    assert np.array_equal(a.parent, b.parent)
    assert np.array_equal(a.taxon, b.taxon)
    assert np.array_equal(a.length.view(np.uint64), b.length.view(np.uint64))
    assert np.array_equal(a.support.view(np.uint64), b.support.view(np.uint64))


def via_objects(path) -> TreeArrays:
    from spectralclustersupertree_amd.load import load_trees

    trees = load_trees(path)
    names = sorted(scs._all_tip_names(trees))
    return TreeArrays.from_trees(trees, [1.0] * len(trees), names)


@pytest.mark.parametrize("name", ["dcm_source_trees.tre", "dcm_iq_source.tre", "supertriplets_source.tre"])
def test_newick_loader_matches_object_loader_on_reference_fixtures(name):
    from tests.reference_cases import DATA_DIR

    path = DATA_DIR / name
    arrays_equal(TreeArrays.from_newick_file(path), via_objects(path))


def test_newick_loader_grammar(tmp_path):
    lines = [
        "((a:0.1,b:2e-3)90:0.5,(c,d)0.75:1,e);",
        "(('it''s a name':1, 'x y' ):3,[comment](b , c)[another]:0.25 , d:);",
        "((a,b),(c,(d,e)label)77);  trailing text is ignored",
        "a;",
        "((a));",
        "(a,b,c,d,e);",
    ]
    path = tmp_path / "t.tre"
    path.write_text("\n".join(lines) + "\n")
    got = TreeArrays.from_newick_file(path)
    arrays_equal(got, via_objects(path))
    assert "it's a name" in got.taxa and "x y" in got.taxa
    # numeric internal labels are supports, others are not
    assert np.nanmax(got.support) == 90.0 and np.count_nonzero(~np.isnan(got.support)) == 3


@pytest.mark.parametrize("seed", range(4))
def test_newick_loader_random_round_trip(tmp_path, seed):
    taxa, trees, weights = random_forest(200 + seed, 50, 12)
    path = tmp_path / "forest.tre"
    with path.open("w") as f:
        for t in trees:
            # supports travel as numeric internal labels, the way the reference's bootstrap
            # inputs carry them
            for node in t.iter_nontips(include_self=True):
                node.name = None if node.support is None else repr(node.support)
            f.write(t.get_newick(with_distances=True, with_node_names=True) + "\n")
    got = TreeArrays.from_newick_file(path, weights)
    want = via_objects(path)
    arrays_equal(got, want)
    assert np.array_equal(got.weights, np.asarray(weights))
    # and the tables they flatten to are the object path's
    trees2 = __import__("spectralclustersupertree_amd.load", fromlist=["load_trees"]).load_trees(path)
    names = sorted(scs._all_tip_names(trees2))
    tables_equal(got.flatten("branch"), fl.flatten_trees(trees2, weights, "branch", names))


@pytest.mark.parametrize("bad", ["((a,b);", "(a,b));", "(a,b),c;", "", "   ", "(a:1x,b);", "('unterminated,b);", "(a,[b);"])
def test_newick_loader_rejects_what_the_object_parser_rejects(tmp_path, bad):
    path = tmp_path / "bad.tre"
    path.write_text("(a,b);\n" + bad + "\n(c,d);\n")
    with pytest.raises(ValueError, match="line 2"):
        TreeArrays.from_newick_file(path)
    with pytest.raises(ValueError):
        make_tree(bad)


def test_construct_supertree_accepts_arrays_and_checks_weights(tmp_path):
    from spectralclustersupertree_amd import construct_supertree

    path = tmp_path / "t.tre"
    path.write_text("(a,(b,c));\n(c,(a,b));\n")
    arrays = TreeArrays.from_newick_file(path)
    with pytest.raises(ValueError, match="must match"):
        construct_supertree(arrays, weights=[1.0])
    with pytest.raises(ValueError, match="Invalid weighting strategy"):
        construct_supertree(arrays, pcg_weighting="nope")


# ---------------------------------------------------------------------------
# contraction groups in C (reference: scs.py:298-334) against the numpy refinement
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_contraction_groups_c_matches_numpy(seed):
    rng = random.Random(300 + seed)
    # forests with real contraction: every tree is a refinement of a few fixed "cherries"
    taxa = [f"t{i:03d}" for i in range(40)]
    cherries = [taxa[i:i + rng.randint(1, 4)] for i in range(0, 40, 4)]
    trees, weights = [], []
    for _ in range(rng.randint(1, 8)):
        picked = [c for c in cherries if rng.random() < 0.8] or cherries[:2]
        units = [TreeNode("", [TreeNode(x) for x in c]) if len(c) > 1 else TreeNode(c[0]) for c in picked]
        rng.shuffle(units)
        while len(units) > 1:
            a, b = units.pop(), units.pop()
            units.append(TreeNode("", [a, b]))
        trees.append(units[0])
        weights.append(1.0)
    names = sorted(scs._all_tip_names(trees))
    tables = fl.flatten_trees(trees, weights, "one", names)
    got, want = fl.contraction_groups(tables), fl.contraction_groups_numpy(tables)
    assert np.array_equal(got, want)
    assert got.max() + 1 <= len(names)
    # and on forests without any structure
    taxa2, trees2, w2 = random_forest(400 + seed, 50, 9, unary=0.0)
    tables2 = fl.flatten_trees(trees2, w2, "one", taxa2)
    assert np.array_equal(fl.contraction_groups(tables2), fl.contraction_groups_numpy(tables2))
    # components: the structured forest is disconnected when few cherries were picked
    assert np.array_equal(fl.pcg_components(tables), fl.pcg_components_numpy(tables))
    assert np.array_equal(fl.pcg_components(tables2), fl.pcg_components_numpy(tables2))


@pytest.mark.parametrize("case", INLINE_CASES, ids=lambda c: c.name)
def test_host_graph_helpers_on_reference_cases(case):
    trees = [make_tree(s) for s in case.trees]
    names = sorted(scs._all_tip_names(trees))
    tables = fl.flatten_trees(trees, [1.0] * len(trees), "one", names)
    assert np.array_equal(fl.pcg_components(tables), fl.pcg_components_numpy(tables))
    assert np.array_equal(fl.contraction_groups(tables), fl.contraction_groups_numpy(tables))


def test_part_without_any_tree_raises_like_the_reference():
    """A part of more than two taxa of which no source tree keeps two: the reference's
    recursive call receives an empty list of induced trees and raises
    (reference: scs.py:63-65 reached from :158) -- no phantom empty leaf in the result."""
    from spectralclustersupertree_amd.treearrays import TreeArrays

    # unary roots: one root side per tree, so the top-level graph is one component
    trees = [make_tree(s) for s in ["(((a,b),(c,p)));", "(((d,b),(c,q)));", "(((e,b),(c,r)));"]]
    taxa = sorted({n for t in trees for n in t.get_tip_names()})
    lone = {taxa.index(x) for x in "ade"}

    def stub(tables, random_state, *, contract_edges):
        # vertices in table order; a, d, e together, everything else on the other side
        members = [np.array([i], dtype=np.int32) for i in range(tables.n_taxa)]
        names = tables.taxa if tables.taxa is not None else taxa
        labels = np.array([1 if names[i] in "ade" else 0 for i in range(tables.n_taxa)])
        return members, labels

    assert len(lone) == 3
    arrays = TreeArrays.from_trees(trees, [1.0, 1.0, 1.0], taxa)
    with pytest.raises(ValueError, match="at least one tree"):
        scs._construct(arrays, "one", True, np.random.RandomState(0), stub)
    with pytest.raises(ValueError, match="at least one tree"):
        scs._construct_objects(trees, [1.0, 1.0, 1.0], "one", True, np.random.RandomState(0), stub)


def test_synthetic_tree_arrays_equal_the_object_path():
    # synthetic.tree_arrays (C, no tree objects) against TreeArrays.from_trees on the same set
    from spectralclustersupertree_amd import synthetic
    from spectralclustersupertree_amd.treearrays import TreeArrays

    for n, m, k in ((40, 5, None), (60, 4, 33), (2, 2, None), (1, 1, None)):
        got = synthetic.tree_arrays(7, n, m, k, random_weights=True)
        trees = synthetic.tree_objects(7, n, m, k)
        names = [synthetic.taxon_name(i) for i in range(n)]
        want = TreeArrays.from_trees(trees, got.weights, names)
        assert np.array_equal(got.node_off, want.node_off)
        assert np.array_equal(got.parent, want.parent)
        assert np.array_equal(got.taxon, want.taxon)
        assert np.array_equal(got.length, want.length, equal_nan=True)
        assert np.array_equal(got.support, want.support, equal_nan=True)
        tabs = synthetic.make_tables(7, n, m, "branch", leaves_per_tree=k, random_weights=True)
        assert np.array_equal(got.weights, tabs.tree_w)
        if n >= 2:
            flat = got.flatten("branch")
            assert np.array_equal(flat.leaf_taxon, tabs.leaf_taxon)
            assert np.array_equal(flat.adj_depth, tabs.adj_depth)
            assert np.array_equal(flat.adj_val, tabs.adj_val)


def test_threaded_host_helpers_equal_serial():
    # libscs_host.so cuts forests above 200 000 nodes over a team of threads: restriction,
    # flattening, present taxa and leaf counts must not depend on the team size
    import hashlib
    import subprocess
    import sys

    code = """
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from spectralclustersupertree_amd import synthetic
a = synthetic.tree_arrays(3, 4000, 70, 3500, random_weights=True)
assert a.parent.size > 200000
keep = np.flatnonzero(np.random.RandomState(0).rand(4000) < 0.6).astype(np.int32)
r = a.restrict(keep)
present = r.present_taxa()
f = r.flatten("branch", local_ids=present)
g = a.flatten("bootstrap")
h = hashlib.md5()
for x in (r.node_off, r.parent, r.taxon, r.length, r.support, r.weights, present, r.leaf_counts(),
          f.tree_off, f.leaf_taxon, f.adj_depth, f.adj_val, g.leaf_taxon, g.adj_depth, g.adj_val):
    h.update(np.ascontiguousarray(x).tobytes())
print(h.hexdigest(), int(f.monotone), int(g.monotone))
""" % str(Path(__file__).resolve().parent.parent)
    out = []
    for threads in ("1", "7"):
        env = dict(os.environ, SCS_HOST_THREADS=threads)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
        out.append(res.stdout.strip())
    assert out[0] == out[1] and len(out[0].split()) == 3


# ---- one-pass multi-way restriction (scs_host_split_*) against the per-part restriction

def _same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.int64) if a.dtype == np.float64 else a,
                                                 b.view(np.int64) if b.dtype == np.float64 else b)


def _check_split(arrays, parts):
    kids = arrays.split(parts)
    assert len(kids) == len(parts)
    for ids, kid in zip(parts, kids):
        ref = arrays.restrict(ids)
        assert kid.n_trees == ref.n_trees
        assert _same_bits(kid.node_off, ref.node_off)
        assert _same_bits(kid.parent, ref.parent)
        mapped = np.where(kid.taxon >= 0, np.asarray(ids, dtype=np.int32)[np.maximum(kid.taxon, 0)], -1)
        assert _same_bits(mapped.astype(np.int32), ref.taxon)
        assert _same_bits(kid.length, ref.length)  # merged lengths: same additions in the same order
        assert _same_bits(kid.support, ref.support)
        assert _same_bits(kid.weights, ref.weights)
        assert np.array_equal(np.asarray(ids)[kid.present_taxa()], ref.present_taxa())
        assert np.array_equal(kid.leaf_counts(), ref.leaf_counts())
        assert all(kid.name(i) == arrays.name(ids[i]) for i in range(len(ids)))


@pytest.mark.parametrize("seed,n,m,k", [(1, 60, 7, 40), (2, 300, 12, 200), (3, 1000, 30, 1000), (4, 5000, 60, 3000)])
def test_split_equals_restrict_per_part(seed, n, m, k):
    # reference: the loop over the parts at scs.py:139-155, each restricted as in :411-455
    from spectralclustersupertree_amd import synthetic

    rs = np.random.RandomState(seed)
    arrays = synthetic.tree_arrays(seed, n, m, k, random_weights=True)
    perm = rs.permutation(n)
    cut = n // 3
    _check_split(arrays, [np.sort(perm[:cut]), np.sort(perm[cut:])])  # a bipartition
    lab = rs.randint(-1, 9, size=n)  # many parts, some taxa in none
    _check_split(arrays, [np.flatnonzero(lab == c) for c in range(9) if np.any(lab == c)])
    _check_split(arrays, [np.array([0, 1, 2]), np.array([5, 9]), np.array([7])])  # tiny parts
    kid = arrays.split([np.sort(perm[: n // 2])])[0]  # a child (taxa renumbered) split again
    sub = rs.permutation(kid.n_taxa)
    _check_split(kid, [np.sort(sub[: kid.n_taxa // 2]), np.sort(sub[kid.n_taxa // 2:])])


def test_split_polytomies_missing_lengths_unary_chains():
    trees = [make_tree(s) for s in [
        "((a:1,b:2,c:3)x:0.5,(d,(e:1,f)y:2)z,(g,h,i,j))", "(((a,b),c),((d,e),(f,(g,(h,(i,j))))))",
        "((a:0.1,(b:0.2,(c:0.3,(d:0.4,e:0.5):0.6):0.7):0.8):0.9,f:1.0)", "(a,b)", "((a,j),(b,i))"]]
    names = sorted({n for t in trees for n in t.get_tip_names()})
    arr = TreeArrays.from_trees(trees, [1.0, 2.0, 0.5, 1.5, 3.0], names)
    for parts in ([[0, 1, 2, 3], [4, 5, 6, 7, 8, 9]], [[0, 9], [1, 8], [2, 3, 4], [5, 6, 7]],
                  [[0, 2, 4, 6, 8], [1, 3, 5, 7, 9]]):
        _check_split(arr, [np.array(p) for p in parts])


def test_split_equals_restrict_on_random_forests_few_and_many_parts():
    # scs_host_split_* reads a part's nodes off per-part bit marks for up to eight parts and sorts
    # event lists above that: both against the per-part restriction, bit for bit, on ragged forests
    from spectralclustersupertree_amd import synthetic

    rs = np.random.RandomState(0)
    checked = 0
    for case in range(250):
        n = int(rs.randint(4, 80))
        m = int(rs.randint(1, 30))
        arrays = synthetic.tree_arrays(case, n, m, int(rs.randint(2, n + 1)))
        k = int(rs.randint(1, 12))
        perm = rs.permutation(n)
        cuts = np.sort(rs.choice(np.arange(1, n), size=min(k - 1, n - 1), replace=False)) if k > 1 else np.array([], int)
        groups = np.split(perm, cuts)
        if rs.rand() < 0.3 and len(groups) > 1:
            groups = groups[:-1]  # some taxa belong to no part
        parts = [np.sort(g).astype(np.int32) for g in groups if len(g)]
        for ids, child in zip(parts, arrays.split(parts)):
            ref = arrays.restrict(ids)
            assert child.n_trees == ref.n_trees and np.array_equal(child.node_off, ref.node_off), (case, len(parts))
            assert np.array_equal(child.parent, ref.parent)
            assert np.array_equal(np.where(child.taxon >= 0, ids[np.clip(child.taxon, 0, None)], -1), ref.taxon)
            assert np.array_equal(child.length, ref.length, equal_nan=True)
            assert np.array_equal(child.support, ref.support, equal_nan=True)
            checked += 1
    assert checked > 1000
