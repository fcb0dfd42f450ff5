"""The recursion's restriction step on the device (``scs_forest_split``, csrc/scs_forest.hip) against
the host sweep of libscs_host.so (``TreeArrays.split`` + ``flatten``), bit for bit -- which
``tests/test_treearrays.py`` in turn holds against the tree-object path (``get_sub_tree`` +
``flatten_trees``; reference: src/sc_supertree/scs.py:139-171, 411-455, 555-564)."""

from __future__ import annotations

import random

import numpy as np
import pytest

from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device
from spectralclustersupertree_amd.treearrays import ResidentArrays, TreeArrays
from tests.test_treearrays import random_forest, tables_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    d = Device(0)
    yield d
    d.close()


@pytest.fixture(autouse=True, params=["thread per tree", "per node"])
def split_kernels(request, monkeypatch):
    """Every test runs through both families of kernels: one thread per tree (small trees; staged in
    LDS where a workgroup's trees fit) and the node-parallel one (big trees) -- forced here on
    forests of any size; a comb deeper than its path buffer falls back to the first family."""
    monkeypatch.setenv("SCS_FOREST_PARALLEL_MIN_TREE_NODES", "1000000000" if request.param == "thread per tree" else "0")
    return request.param


def _same_bits(a: np.ndarray, b: np.ndarray) -> bool:
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def forests_equal(host: TreeArrays, res: ResidentArrays) -> None:
    got = res.to_host()
    assert got.n_taxa == host.n_taxa and got.n_trees == host.n_trees
    assert np.array_equal(got.node_off, host.node_off)
    assert np.array_equal(got.parent, host.parent)
    assert np.array_equal(got.taxon, host.taxon)
    assert _same_bits(got.length, host.length)  # NaN = None, signed zeros: bit for bit
    assert _same_bits(got.support, host.support)
    assert _same_bits(got.weights, host.weights)
    assert np.array_equal(res.present_taxa(), host.present_taxa())
    assert np.array_equal(res.leaf_counts(), host.leaf_counts())
    if host.ids is None:
        assert res.ids is None
    else:
        assert np.array_equal(res.ids, host.ids)


def random_parts(rng: random.Random, n_taxa: int, n_parts: int, drop: float):
    ids = list(range(n_taxa))
    rng.shuffle(ids)
    ids = ids[: max(n_parts * 2, int(n_taxa * (1.0 - drop)))]
    cuts = sorted(rng.sample(range(1, len(ids)), n_parts - 1)) if n_parts > 1 else []
    parts, lo = [], 0
    for hi in [*cuts, len(ids)]:
        parts.append(np.asarray(sorted(ids[lo:hi]), dtype=np.int32))
        lo = hi
    return parts


def compare_split(dev, arrays: TreeArrays, parts, strategy: str, levels: int = 2, rng=None) -> None:
    """Children of the device split against the host's: node arrays, present taxa, tables; then
    (``levels``) the children split again, resident forest against host forest."""
    res = ResidentArrays.from_host(arrays, dev)
    frontier = [(arrays, res, parts)]
    for _ in range(levels):
        nxt = []
        for host, resident, pp in frontier:
            want = host.split(pp)
            got = resident.split(pp, strategy)
            assert len(got) == len(want)
            for w, g in zip(want, got):
                assert isinstance(g, ResidentArrays)
                forests_equal(w, g)
                if w.n_trees:
                    present = w.present_taxa()
                    tables_equal(g.flatten(strategy, local_ids=present), w.flatten(strategy, local_ids=present))
                    tables_equal(g.flatten(strategy), w.flatten(strategy))
                    if rng is not None and w.n_taxa >= 4:
                        nxt.append((w, g, random_parts(rng, w.n_taxa, 2, 0.1)))
        frontier = nxt


@pytest.mark.parametrize("strategy", ["one", "depth", "branch"])
@pytest.mark.parametrize("seed", range(5))
def test_device_split_matches_host_split_on_random_forests(dev, monkeypatch, seed, strategy):
    # multifurcations, unary chains, missing lengths, negative lengths (monotone flag), partial
    # coverage, taxa of no part, two to eight parts, two levels
    monkeypatch.setenv("SCS_DEVICE_SPLIT_MIN_NODES", "0")
    rng = random.Random(100 + seed)
    taxa, trees, weights = random_forest(seed, 60, 25, neg_len=0.1 if seed % 2 else 0.0)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    for n_parts in (2, 3, 8):
        compare_split(dev, arrays, random_parts(rng, 60, n_parts, 0.2), strategy, rng=rng)


def test_device_split_bootstrap_and_missing_support(dev, monkeypatch):
    monkeypatch.setenv("SCS_DEVICE_SPLIT_MIN_NODES", "0")
    rng = random.Random(7)
    taxa, trees, weights = random_forest(3, 50, 12, none_sup=0.0, unary=0.0)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    compare_split(dev, arrays, random_parts(rng, 50, 2, 0.0), "bootstrap", rng=rng)
    # an inner node without support: the reference fails in `length * tree_weight` (scs.py:656)
    taxa, trees, weights = random_forest(4, 30, 6, none_sup=1.0, unary=0.0)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    res = ResidentArrays.from_host(arrays, dev)
    with pytest.raises(TypeError):
        res.split(random_parts(rng, 30, 2, 0.0), "bootstrap")


def test_device_split_of_full_binary_forests_with_weights(dev, monkeypatch):
    # the benchmark's shape in small: every tree over all taxa, per-tree weights, a few levels deep
    monkeypatch.setenv("SCS_DEVICE_SPLIT_MIN_NODES", "0")
    arrays = synthetic.tree_arrays(5, 300, 40)
    arrays.weights[:] = np.random.RandomState(1).uniform(0.5, 2.0, arrays.n_trees)
    rng = random.Random(2)
    compare_split(dev, arrays, random_parts(rng, 300, 2, 0.0), "branch", levels=4, rng=rng)


def test_device_split_deep_caterpillars_and_single_leaf_parts(dev, monkeypatch):
    # a comb of 3 000 leaves (stack depth = tree height), a part that keeps ONE leaf of a tree
    # (dropped there), a part no tree keeps two leaves of (no trees at all)
    monkeypatch.setenv("SCS_DEVICE_SPLIT_MIN_NODES", "0")
    from spectralclustersupertree_amd.tree import TreeNode

    n = 3000
    names = [f"t{i:05d}" for i in range(n)]
    node = TreeNode(names[0], None, 0.5)
    for i in range(1, n):
        node = TreeNode("", [node, TreeNode(names[i], None, 0.25 + i * 1e-3)], 0.125 + i * 1e-4, 80.0)
    node.length = None
    small = TreeNode("", [TreeNode(names[5], None, 1.0), TreeNode(names[2000], None, 2.0)], None, None)
    arrays = TreeArrays.from_trees([node, small], [1.0, 3.0], names)
    parts = [np.arange(0, 1500, dtype=np.int32), np.arange(1500, 2999, dtype=np.int32),
             np.asarray([2999], dtype=np.int32)]
    compare_split(dev, arrays, parts, "branch", levels=1)
    compare_split(dev, arrays, parts, "depth", levels=1)


def test_whole_recursion_with_the_split_on_the_device(monkeypatch):
    # the same supertree, label for label, whether the forests are split on the host or on the device
    from spectralclustersupertree_amd import scs

    arrays = synthetic.tree_arrays(11, 700, 60)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SCS_DEVICE_SPLIT", mode)
        monkeypatch.setenv("SCS_DEVICE_SPLIT_MIN_NODES", "0")
        rs = np.random.RandomState(5)
        tree = scs.construct_supertree(synthetic.tree_arrays(11, 700, 60), pcg_weighting="branch", random_state=rs)
        out[mode] = (tree.get_newick(), rs.randint(1 << 30))
    assert out["0"] == out["1"]
    assert arrays.n_trees == 60


def test_small_solve_from_resident_tables_equals_the_host_packed_one(dev, monkeypatch):
    # a child of the device split is solved straight from its device tables (k_small_pack: present-taxon
    # and contraction renumbering applied on the device) -- W, eigenvalues and embedding bit for bit
    # those of the batch packed on the host; few trees: taxa contract, some are absent
    from spectralclustersupertree_amd import scs

    monkeypatch.setenv("SCS_DEVICE_SPLIT_MIN_NODES", "0")
    rng = random.Random(12)
    for seed, n, m in ((3, 90, 6), (4, 160, 40), (5, 40, 3)):
        taxa, trees, weights = random_forest(seed, n, m, none_sup=0.0, unary=0.05)
        arrays = TreeArrays.from_trees(trees, weights, taxa)
        res = ResidentArrays.from_host(arrays, dev)
        for child in res.split(random_parts(rng, n, 2, 0.05), "branch"):
            if child.n_trees < 1 or len(child.present_taxa()) < 3:
                continue
            tables = child.flatten("branch", local_ids=child.present_taxa())
            assert tables.resident is not None and tables.resident[0] is child.forest
            work, perm, group_start, n_groups = scs.prepare_node(tables, True)
            if n_groups < 2 or work.n_taxa > dev.SMALL_MAX_TAXA:
                continue
            assert work.resident is not None
            got = {}
            for mode in ("1", "0"):
                monkeypatch.setenv("SCS_RESIDENT_SOLVE", mode)
                got[mode] = dev.small_solve([(work, group_start)], want_w=True)[0]
            for a, b in zip(got["1"], got["0"]):
                assert np.array_equal(a, b)


def test_tables_of_a_larger_node_straight_from_the_resident_forest(dev, monkeypatch):
    # scs_tables_from_forest: device-to-device copies + the renumbering kernel instead of the host -> HBM
    # upload; the graph built from them is the one built from the uploaded host tables, bit for bit --
    # also from a second context on the same GPU (the look-ahead worker's)
    from spectralclustersupertree_amd import scs

    monkeypatch.setenv("SCS_DEVICE_SPLIT_MIN_NODES", "0")
    rng = random.Random(3)
    taxa, trees, weights = random_forest(8, 400, 30, none_sup=0.0, unary=0.05)
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    res = ResidentArrays.from_host(arrays, dev)
    other = Device(0)
    try:
        for child in res.split(random_parts(rng, 400, 2, 0.05), "branch"):
            tables = child.flatten("branch", local_ids=child.present_taxa())
            work, perm, group_start, n_groups = scs.prepare_node(tables, True)
            assert work.resident is not None
            got = {}
            for mode, d in (("1", dev), ("0", dev), ("1", other)):
                monkeypatch.setenv("SCS_RESIDENT_SOLVE", mode)
                dtab = d.upload(work)
                g = dtab.build()
                got[(mode, d is dev)] = g.download()
                g.free()
                dtab.free()
            assert np.array_equal(got[("1", True)], got[("0", True)])
            assert np.array_equal(got[("1", False)], got[("0", True)])
    finally:
        other.close()


def test_upload_refuses_malformed_arrays(dev):
    """``scs_forest_upload`` checks the arrays where they arrive (round 6): every later kernel walks ``parent``
    upwards and indexes by ``taxon`` without looking again -- a non-root without an earlier parent would walk out of
    its tree or never stop (the node-parallel family only flagged it after the whole pipeline had run)."""
    from spectralclustersupertree_amd import _native as nv
    from spectralclustersupertree_amd.backend import DeviceForest

    arrays = synthetic.tree_arrays(3, 40, 6)
    good = (np.ascontiguousarray(arrays.node_off), arrays.parent.copy(), arrays.taxon.copy(), arrays.length.copy(),
            arrays.support.copy(), arrays.weights.copy())
    n_leaves = int(np.count_nonzero(arrays.taxon >= 0))

    def upload(parent=None, taxon=None):
        f = DeviceForest.upload(dev, arrays.n_taxa, good[0], good[1] if parent is None else parent,
                                good[2] if taxon is None else taxon, good[3], good[4], good[5], n_leaves)
        f.free()

    upload()  # the arrays as they are: fine
    for where, value in ((5, 5), (5, 7), (3, -1), (int(arrays.node_off[2]), 0)):
        bad = good[1].copy()
        bad[where] = value  # its own parent / a later node / a second root / a root with a parent
        with pytest.raises(nv.ScsError) as err:
            upload(parent=bad)
        assert err.value.code == nv.EINVAL and "malformed" in str(err.value)
    for value in (arrays.n_taxa, -2):
        bad = good[2].copy()
        bad[np.flatnonzero(bad >= 0)[0]] = value
        with pytest.raises(nv.ScsError):
            upload(taxon=bad)
