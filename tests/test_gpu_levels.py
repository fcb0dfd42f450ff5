"""The level-synchronous recursion (``spectralclustersupertree_amd/levels.py``, ``scs_forest_split_level``,
``scs_small_solve_begin_level``; reference: src/sc_supertree/scs.py:122-171 -- components, contraction, the
spectral split, restriction to each part, the recursive call with the SAME RandomState).

Held against three things:
* the oracle, node by node, with the engine forced onto every subtree (``compare_with_oracle`` of
  ``tests/test_gpu_recursion.py``: same spectral calls, same vertices, identical labels, same stream position);
* the node-by-node walk of the product itself (same Newick string, same trace, same stream position) -- also
  with provisional labels that are deliberately WRONG (the verification must repair every one of them) and
  with the device refusing the redo (the host-array fallback);
* the host routines for what the level split reports from the device: the restricted forests and their
  tables (``TreeArrays.split`` + ``flatten``), the components (``flatten.pcg_components``) and -- via the
  signatures -- the contraction groups (``flatten.contraction_groups``).
"""

from __future__ import annotations

import random

import numpy as np
import pytest

from spectralclustersupertree_amd import flatten as fl
from spectralclustersupertree_amd import levels, scs, synthetic
from spectralclustersupertree_amd.backend import Device
from spectralclustersupertree_amd.scs import trace_nodes
from spectralclustersupertree_amd.treearrays import _STRATEGY_CODE, ResidentArrays, TreeArrays
from tests.test_gpu_recursion import compare_with_oracle, recursion_input
from tests.test_treearrays import random_forest, tables_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    d = Device(0)
    yield d
    d.close()


@pytest.fixture()
def engine_everywhere(monkeypatch):
    """Every child subtree goes through the engine, whatever its size."""
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "1000000")
    monkeypatch.setenv("SCS_SPEC_MIN_NODES", "0")


def _run(arrays, strategy, contract, seed):
    rs = np.random.RandomState(seed)
    with trace_nodes() as trace:
        tree = scs._construct(arrays, strategy, contract, rs)
    return tree.get_newick(), int(rs.randint(1 << 30)), [(e["vertices"], e["labels"].tolist()) for e in trace]


# ---------------------------------------------------------------------------------------------- vs the oracle
def test_engine_against_the_oracle_branch_partial_coverage_twins(engine_everywhere):
    trees, weights = recursion_input(5, 900, 30, 600, 80, weighted=True)
    trace, ties = compare_with_oracle(trees, weights, "branch", seed=3)
    assert not ties
    assert levels.stats["roots"] >= 1 and levels.stats["nodes"] >= len(trace) // 2
    assert any(len(v) > 1 for e in trace for v in e["vertices"])  # contraction inside the engine
    assert levels.stats["exact_group_nodes"] >= 1  # ... found through colliding signatures


def test_engine_against_the_oracle_bootstrap(engine_everywhere):
    trees, weights = recursion_input(9, 500, 25, 350, 40, weighted=False)
    _, ties = compare_with_oracle(trees, weights, "bootstrap", seed=11)
    assert not ties


@pytest.mark.parametrize("strategy", ["one", "depth"])
def test_engine_against_the_oracle_integer_strategies(engine_everywhere, strategy):
    trees, weights = recursion_input(2, 400, 16, 300, 30, weighted=True)
    trace, ties = compare_with_oracle(trees, weights, strategy, seed=1, as_arrays=True, ties_allowed=True)
    assert len(ties) <= len(trace) // 10


def test_engine_against_the_oracle_without_contraction(engine_everywhere):
    trees, weights = recursion_input(4, 500, 20, 350, 50, weighted=False)
    trace, ties = compare_with_oracle(trees, weights, "branch", seed=7, contract_edges=False)
    assert not ties
    assert all(len(v) == 1 for e in trace for v in e["vertices"])


# ------------------------------------------------------------------------------- vs the node-by-node walk
CASES = [(60, 10, 40, "branch", True), (200, 30, 120, "depth", True), (700, 40, None, "one", False),
         (1200, 60, 800, "bootstrap", True), (2500, 30, None, "branch", True)]


@pytest.mark.parametrize("from_parts", [True, False])
@pytest.mark.parametrize("n, m, leaves, strategy, contract", CASES)
def test_engine_equals_the_node_by_node_walk(monkeypatch, n, m, leaves, strategy, contract, from_parts):
    """``from_parts``: the engine starts at the root's own forest and parts (levels.construct_parts, the
    default) or at every child after the node-by-node split (levels.construct: what a node of more than
    eight parts still does)."""
    kw = {} if leaves is None else {"leaves_per_tree": leaves}
    arrays = synthetic.tree_arrays(n + m, n, m, random_weights=True, **kw)
    monkeypatch.setenv("SCS_SPEC_MIN_NODES", "0")
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "1000000")
    monkeypatch.setenv("SCS_SPEC_FROM_PARTS", "1" if from_parts else "0")
    with_engine = _run(arrays, strategy, contract, 5)
    st = dict(levels.stats)
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "0")
    node_by_node = _run(arrays, strategy, contract, 5)
    assert levels.stats["roots"] == 0 and st["roots"] >= 1
    assert (st["from_parts"] >= 1) == from_parts
    assert with_engine == node_by_node


@pytest.mark.parametrize("n, m, leaves, strategy, contract", [(700, 40, None, "one", False), (1200, 60, 800, "bootstrap", True),
                                                              (2500, 30, None, "branch", True)])
def test_nodes_above_the_cap_are_embedded_behind_the_walk(monkeypatch, n, m, leaves, strategy, contract):
    """Nodes above the cap are deferred (nothing computed below them on a guess), so their level does not wait for
    their embeddings: ONE job of the look-ahead queue takes them in the order of the visit and the walk picks each up
    when it arrives (``levels._Lazy``).  Only WHEN a node is embedded changes -- the tree, the trace of every spectral
    call and the stream position are those of the node-by-node walk."""
    kw = {} if leaves is None else {"leaves_per_tree": leaves}
    arrays = synthetic.tree_arrays(n + m, n, m, random_weights=True, **kw)
    monkeypatch.setenv("SCS_SPEC_MIN_NODES", "0")
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "150")
    with_engine = _run(arrays, strategy, contract, 5)
    st = dict(levels.stats)
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "0")
    node_by_node = _run(arrays, strategy, contract, 5)
    assert st["roots"] >= 1 and st["lazy_nodes"] >= 3 and st["deferred"] >= st["lazy_nodes"]
    assert with_engine == node_by_node


@pytest.mark.parametrize("global_sig", [False, True])
def test_large_universes_with_twins_and_partial_coverage(monkeypatch, global_sig):
    """Levels of more than 2 048 ids take the analysis kernels of large universes: the union-find with several
    leaves per thread, and the signatures folded per tile of taxa in LDS (``global_sig``: the fall-back that adds
    them with global atomics, forced).  300 twinned taxa must be contracted, partial coverage leaves components:
    the engine's recursion equals the node-by-node walk, trace and stream position included."""
    trees, weights = recursion_input(17, 2600, 14, 2000, 300, weighted=True)
    names = sorted({tip for t in trees for tip in t.get_tip_names()})
    arrays = TreeArrays.from_trees(trees, weights, names)
    monkeypatch.setenv("SCS_SPEC_MIN_NODES", "0")
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "1000000")
    if global_sig:
        monkeypatch.setenv("SCS_ANALYZE_GLOBAL_SIG", "1")
    with_engine = _run(arrays, "branch", True, 9)
    st = dict(levels.stats)
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "0")
    node_by_node = _run(arrays, "branch", True, 9)
    assert st["roots"] >= 1 and st["exact_group_nodes"] >= 1  # (equal signatures found, the exact routine asked)
    assert any(len(v) > 1 for v, _ in with_engine[2])  # contraction happened
    assert with_engine == node_by_node


def test_wrong_provisional_labels_are_all_repaired(monkeypatch, engine_everywhere):
    """Provisional labels that are wrong on purpose -- one vertex of every third node moved to the other part,
    every fifth node's labels reversed -- change nothing: the walk assigns every node's labels with the
    caller's stream and redoes what the provisional partition got wrong (the planted 'tie')."""
    from spectralclustersupertree_amd import kmeans2

    arrays = synthetic.tree_arrays(77, 600, 40, random_weights=True)
    monkeypatch.setenv("SCS_SPEC_VOTES", "1")  # (one label assignment per node: the spoiled one is what the engine bets on)
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "0")
    want = _run(arrays, "branch", True, 9)
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "1000000")
    real = kmeans2.provisional_labels
    count = {"nodes": 0}

    def spoiled(maps, vptr, rs):
        lab = real(maps, vptr, rs)
        if lab is None:
            return None
        for k in range(len(vptr) - 1):
            a, b = int(vptr[k]), int(vptr[k + 1])
            if b - a < 2:
                continue
            count["nodes"] += 1
            if count["nodes"] % 3 == 0 and b - a >= 4:
                lab[a] = 1 - lab[a]
            if count["nodes"] % 5 == 0:
                lab[a:b] = 1 - lab[a:b]
        return lab

    monkeypatch.setattr(kmeans2, "provisional_labels", spoiled)
    got = _run(arrays, "branch", True, 9)
    assert got == want
    assert levels.stats["mismatches"] >= 10  # the planted ones were met, and repaired
    # ... and once more with the device refusing every redo: the node's forest comes back as host arrays
    from spectralclustersupertree_amd import _native as nv

    def refuse(self, lev, k, labels):
        raise nv.ScsError(nv.ENOMEM, "planted")

    monkeypatch.setattr(levels.Engine, "redo", refuse)
    got = _run(arrays, "branch", True, 9)
    assert got == want
    assert levels.stats["fallbacks"] >= 10


def test_nodes_on_which_the_votes_disagree_are_deferred_not_guessed(monkeypatch, engine_everywhere):
    """Three label assignments per node from different draws: where they do not agree the engine computes
    nothing below the node (no work to throw away) and the walk takes the subtree up again with the labels of
    record -- the same supertree, labels and stream position as with one vote, and as node by node."""
    arrays = synthetic.tree_arrays(123, 1500, 30, random_weights=True)
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "0")
    want = _run(arrays, "branch", True, 2)
    monkeypatch.setenv("SCS_SPEC_MAX_TAXA", "1000000")
    got = {}
    for votes in ("1", "3", "5"):
        monkeypatch.setenv("SCS_SPEC_VOTES", votes)
        monkeypatch.setenv("SCS_SPEC_DEFER_MIN", "0")
        got[votes] = (_run(arrays, "branch", True, 2), dict(levels.stats))
        assert got[votes][0] == want
    assert got["1"][1]["deferred"] == 0
    assert got["3"][1]["deferred"] >= 1 and got["5"][1]["deferred"] >= got["3"][1]["deferred"]
    # fewer bets lost than with one vote, and fewer nodes computed in vain
    assert got["5"][1]["mismatches"] <= got["1"][1]["mismatches"]


def test_a_larger_node_that_fails_on_a_worker_is_solved_on_the_engines_own_context(monkeypatch, engine_everywhere):
    """The larger nodes of a level go to the look-ahead workers' contexts; a job that fails there (two nodes in
    flight need more device memory than one) is solved again on the engine's own context: the same result."""
    import threading

    arrays = synthetic.tree_arrays(31, 1800, 24, random_weights=True)
    want = _run(arrays, "branch", True, 4)
    real, failed = levels.Engine._large_job, []

    def flaky(self, lev, k, relabel, gs_patch):
        job = real(self, lev, k, relabel, gs_patch)

        def run(dev):
            if threading.current_thread().name.startswith("scs-ahead"):
                failed.append(k)
                msg = "libscs_hip error -3: out of device memory (simulated)"
                raise RuntimeError(msg)
            return job(dev)

        return run

    monkeypatch.setattr(levels.Engine, "_large_job", flaky)
    assert _run(arrays, "branch", True, 4) == want
    assert failed


def test_forests_that_fall_apart_and_single_tree_nodes(engine_everywhere):
    """Partial coverage by few trees: components instead of spectral calls, nodes left with ONE tree (grafted
    as it is, scs.py:96-98), taxa no surviving tree holds (scs.py:166-170) -- all inside the engine."""
    for seed in range(6):
        taxa, trees, weights = random_forest(100 + seed, 90, 4 + seed)
        arrays = TreeArrays.from_trees(trees, weights, taxa)
        for strategy in ("branch", "depth"):
            monkey_off = _run_with(arrays, strategy, "0", seed)
            monkey_on = _run_with(arrays, strategy, "1000000", seed)
            assert monkey_on == monkey_off


def _run_with(arrays, strategy, cap, seed):
    import os

    old = os.environ.get("SCS_SPEC_MAX_TAXA")
    os.environ["SCS_SPEC_MAX_TAXA"] = cap
    try:
        try:
            return _run(arrays, strategy, True, seed)
        except ValueError as exc:  # (an empty induced forest raises on both paths, scs.py:63-65)
            return ("ValueError", str(exc))
    finally:
        if old is None:
            del os.environ["SCS_SPEC_MAX_TAXA"]
        else:
            os.environ["SCS_SPEC_MAX_TAXA"] = old


# ------------------------------------------------------------------------- the level split against the host
def _level_of(rng, arrays: TreeArrays, n_nodes: int):
    """``n_nodes`` disjoint taxon sets of ``arrays`` as the nodes of one level: the host's split gives the
    node forests, their concatenation (taxa renumbered to consecutive ranges) is the level forest."""
    ids = list(range(arrays.n_taxa))
    rng.shuffle(ids)
    cuts = sorted(rng.sample(range(1, len(ids)), n_nodes - 1))
    sets = [np.asarray(sorted(ids[a:b]), dtype=np.int32) for a, b in zip([0, *cuts], [*cuts, len(ids)])]
    sets = [s for s in sets if len(s) >= 3]
    nodes = [c for c in arrays.split(sets) if c.n_trees >= 1]
    return nodes


@pytest.mark.parametrize("family", ["thread per tree", "per node"])
@pytest.mark.parametrize("strategy", ["branch", "depth", "one"])
def test_level_split_equals_the_host_split_of_every_node(dev, monkeypatch, strategy, family):
    monkeypatch.setenv("SCS_FOREST_PARALLEL_MIN_TREE_NODES", "1000000000" if family == "thread per tree" else "0")
    rng = random.Random(11)
    taxa, trees, weights = random_forest(11, 160, 30, none_sup=0.0)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    nodes = _level_of(rng, arrays, 6)
    # ---- the level forest: node after node, taxa as consecutive ranges
    u_sz = np.asarray([c.n_taxa for c in nodes], dtype=np.int32)
    u_lo = np.cumsum(u_sz) - u_sz
    node_off, parent, taxon, length, support, weights, t_end = [0], [], [], [], [], [], []
    for c, lo in zip(nodes, u_lo):
        node_off.extend((c.node_off[1:] + node_off[-1]).tolist())
        parent.append(c.parent)
        taxon.append(np.where(c.taxon >= 0, c.taxon + lo, -1).astype(np.int32))
        length.append(c.length)
        support.append(c.support)
        weights.append(c.weights)
        t_end.append(len(node_off) - 1)
    level = TreeArrays(n_taxa=int(u_sz.sum()), node_off=np.asarray(node_off, dtype=np.int64),
                       parent=np.concatenate(parent), taxon=np.concatenate(taxon), length=np.concatenate(length),
                       support=np.concatenate(support), weights=np.concatenate(weights), taxa=arrays.taxa)
    forest = ResidentArrays.from_host(level, dev).forest
    # ---- every node split in two or three parts (some taxa in none)
    part_of = np.full(level.n_taxa, -1, dtype=np.int32)
    new_id = np.zeros(level.n_taxa, dtype=np.int32)
    parts_of_node, order = [], []
    for k, c in enumerate(nodes):
        n_parts = 2 + (k % 2)
        local = list(range(c.n_taxa))
        rng.shuffle(local)
        local = local[: max(6, int(0.9 * len(local)))]
        cuts = sorted(rng.sample(range(1, len(local)), n_parts - 1))
        sets = [np.asarray(sorted(local[a:b]), dtype=np.int32) for a, b in zip([0, *cuts], [*cuts, len(local)])]
        parts_of_node.append(sets)
    n_parts = max(len(s) for s in parts_of_node)
    at = 0
    child_base = {}
    for b in range(n_parts):  # the children's numbering: part-major, node order -- the union's tree order
        for k, sets in enumerate(parts_of_node):
            if b < len(sets):
                ids = sets[b] + u_lo[k]
                part_of[ids] = b
                new_id[ids] = at + np.arange(len(ids), dtype=np.int32)
                child_base[(b, k)] = at
                order.append((b, k))
                at += len(ids)
    union, child_trees, child_leaves, present, comp_root, sig = forest.split_level(
        part_of, new_id, n_parts, at, _STRATEGY_CODE[strategy], np.asarray(t_end, dtype=np.int32))
    t_at = 0
    for b, k in order:
        want = nodes[k].split([parts_of_node[k][b]])[0]
        m = int(child_trees[b, k])
        assert m == want.n_trees
        base, size = child_base[(b, k)], len(parts_of_node[k][b])
        if m:
            node_off_g, parent_g, taxon_g, length_g, support_g, weights_g = union.download(t_at, t_at + m)
            assert np.array_equal(node_off_g, want.node_off) and np.array_equal(parent_g, want.parent)
            assert np.array_equal(np.where(taxon_g >= 0, taxon_g - base, -1), want.taxon)
            assert np.array_equal(length_g.view(np.uint64), want.length.view(np.uint64))
            assert np.array_equal(support_g.view(np.uint64), want.support.view(np.uint64))
            assert np.array_equal(weights_g.view(np.uint64), want.weights.view(np.uint64))
            tree_off, leaf_taxon, adj_depth, adj_val, tree_w = union.tables_range(t_at, t_at + m)
            tab = want.flatten(strategy)
            assert int(child_leaves[b, k]) == tab.n_leaves
            got = fl.TreeTables(n_taxa=size, tree_off=tree_off, leaf_taxon=(leaf_taxon - base).astype(np.int32),
                                adj_depth=adj_depth, adj_val=adj_val, tree_w=tree_w, monotone=tab.monotone)
            tables_equal(got, tab)
            here = present[base:base + size].astype(bool)
            assert np.array_equal(np.flatnonzero(here), want.present_taxa())
            # ---- what the device reports about the child's graph, against the host routines
            local = want.flatten(strategy, local_ids=want.present_taxa())
            comp = fl.pcg_components(local)
            roots = comp_root[base:base + size][here] - base
            first_of = {}
            for i, c in enumerate(comp):
                first_of.setdefault(int(c), int(want.present_taxa()[i]))
            assert [first_of[int(c)] for c in comp] == roots.tolist()
            groups = fl.contraction_groups(local)
            s = sig[base:base + size][here]
            for i in range(len(groups)):
                for j in range(i + 1, len(groups)):
                    if groups[i] == groups[j]:  # contracted taxa carry equal signatures (the converse: exact path)
                        assert s[i, 0] == s[j, 0] and s[i, 1] == s[j, 1]
        t_at += m
    assert t_at == union.n_trees
