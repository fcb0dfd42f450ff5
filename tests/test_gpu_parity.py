"""Parity of the HIP path against the oracle, through the C-ABI (needs an MI355X).

Bars: W (integer/rounded-sum work) bit-exact; Fiedler column within 1e-10 of
scikit-learn's embedding (BASELINE.json north_star); labels identical.
"""

import threading

import numpy as np
import pytest
from reference_cases import DATA_DIR, FILE_CASES, INLINE_CASES, NOT_COMPLETED_CASE

from oracle import scs_oracle as so
from oracle import tables_oracle as to
from spectralclustersupertree_amd import _native as nv
from spectralclustersupertree_amd import flatten as fl
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device
from spectralclustersupertree_amd.scs import relabel_for_contraction
from spectralclustersupertree_amd.tree import NotCompleted, load_tree, make_tree

pytestmark = pytest.mark.gpu

def fl_leaf(name):
    from spectralclustersupertree_amd.tree import TreeNode

    return TreeNode(name, None, 0.05)


def fl_tree(children):
    from spectralclustersupertree_amd.tree import TreeNode

    return TreeNode("", children)


FIEDLER_TOL = 1e-10  # north_star: Fiedler-vector entries within 1e-10 fp64


@pytest.fixture(scope="module")
def dev():
    d = Device(0)
    yield d
    d.close()


# ---------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 24, 33, 48, 63, 64])
def test_jacobi_matches_lapack(dev, n):
    rs = np.random.RandomState(n)
    a = rs.standard_normal((n, n))
    a = a + a.T
    w, v = dev.debug_jacobi(a)
    w_ref = np.linalg.eigvalsh(a)[::-1]
    scale = max(1.0, np.abs(w_ref).max())
    assert np.max(np.abs(w - w_ref)) <= 1e-12 * scale
    assert np.max(np.abs(a @ v - v * w)) <= 1e-11 * scale
    assert np.max(np.abs(v.T @ v - np.eye(n))) <= 1e-12


@pytest.mark.parametrize("use_mfma", [False, True], ids=["valu", "mfma_f64"])
@pytest.mark.parametrize(("n", "ka", "kb"), [(5, 1, 8), (257, 8, 8), (1000, 16, 24), (4099, 24, 24), (3000, 48, 48), (777, 32, 16)])
def test_gram_kernels(dev, use_mfma, n, ka, kb):
    rs = np.random.RandomState(ka * 100 + kb)
    a = rs.standard_normal((n, ka))
    b = rs.standard_normal((n, kb)) + 0.25  # asymmetric on purpose
    got = dev.debug_gram(a, b, use_mfma)
    ref = a.T @ b
    assert np.max(np.abs(got - ref)) <= 1e-10 * np.sqrt(n)


# ---------------------------------------------------------------------------
# proper cluster graph
# ---------------------------------------------------------------------------
def _build_and_compare(dev, tables, row_ranges=None, expect_spec=False):
    w_ref, _ = to.pcg_dense(tables)
    dtab = dev.upload(tables)
    try:
        for rb, re_ in row_ranges or [(0, tables.n_taxa)]:
            g = dtab.build(rb, re_)
            w = g.download()
            if expect_spec:  # every tree batch went through k_accumulate_spec (SCS_WIDE=1 lifts the gates)
                assert g.build_stats["spec_batches"] == g.build_stats["n_batches"] >= 1, g.build_stats
            g.free()
            assert w.shape == (re_ - rb, tables.n_taxa)
            diff = w != w_ref[rb:re_]
            assert not diff.any(), (
                f"{int(diff.sum())} cells differ (rows {rb}:{re_}); first at {np.argwhere(diff)[:5].tolist()}, "
                f"got {w[diff][:5]}, want {w_ref[rb:re_][diff][:5]}"
            )
    finally:
        dtab.free()
    return w_ref


@pytest.mark.parametrize("strategy", ["one", "depth", "branch", "bootstrap"])
# (the last case has more trees than a batch of a large problem takes, 256; a problem this small
# walks them in one batch -- several batches: test_build_batched_matches_single_batch and the
# configs[2]-size tests)
@pytest.mark.parametrize(("n", "m", "k"), [(37, 6, 20), (64, 5, 64), (100, 9, 71), (300, 12, 300), (700, 7, 512),
                                           (330, 600, 200)])
def test_build_bit_exact_synthetic(dev, strategy, n, m, k):
    tables = synthetic.make_tables(100 + n, n, m, strategy, leaves_per_tree=k, random_weights=(n % 2 == 0))
    _build_and_compare(dev, tables)


def test_build_row_blocks_bit_exact(dev):
    # the non-symmetric schedule used by row-partitioned ranks, ragged boundaries
    tables = synthetic.make_tables(5, 333, 10, "branch", leaves_per_tree=250, random_weights=True)
    _build_and_compare(dev, tables, [(0, 100), (100, 333), (64, 65), (7, 201)])


def test_build_monotone_and_general_kernels_agree(dev, monkeypatch):
    # the monotone fast path must give the bits of the general kernel (and of the oracle)
    tables = synthetic.make_tables(31, 700, 15, "branch", leaves_per_tree=600, random_weights=True)
    assert tables.monotone
    w_fast = _build_and_compare(dev, tables)
    tables.monotone = False
    w_general = _build_and_compare(dev, tables)
    assert np.array_equal(w_fast, w_general)


def test_build_negative_lengths_and_weights_take_general_kernel(dev):
    trees = [make_tree(s) for s in ["((a:1,b:1):-0.5,((c:1,d:1):2,e:1):0.25)", "(((a:1,c:1):1,b:1):1,(d:1,e:1):1)"]]
    names = sorted(so._all_tips(trees))
    t1 = fl.flatten_trees(trees, [1.0, 2.0], "branch", names)
    assert not t1.monotone
    _build_and_compare(dev, t1)
    t2 = fl.flatten_trees(trees[1:], [-1.5], "depth", names)
    assert not t2.monotone
    _build_and_compare(dev, t2)


def test_build_general_kernel_deep_caterpillars(dev):
    # values that go up and down along a root path (negative lengths -> the general tile kernel,
    # scs_gen.h) on trees as deep as they are wide: long runs of rows that share the column's
    # LCA, depths far beyond the 63 gaps of a tile, taxa missing from some trees
    rs = np.random.RandomState(5)
    taxa = [f"t{i}" for i in range(330)]
    trees = []
    for k in range(7):
        order = list(rs.permutation(taxa)[: 330 - 40 * (k % 3)])
        nwk = f"{order[0]}:1"
        for name in order[1:]:
            left = rs.rand() < 0.5
            ln = float(np.round(rs.uniform(-1.0, 2.0), 3))
            nwk = f"({name}:1,{nwk}):{ln}" if left else f"({nwk},{name}:1):{ln}"
        trees.append(make_tree(nwk + ";"))
    names = sorted(so._all_tips(trees))
    tables = fl.flatten_trees(trees, list(rs.uniform(0.5, 2.0, len(trees))), "branch", names)
    assert not tables.monotone
    _build_and_compare(dev, tables)
    _build_and_compare(dev, tables, [(0, 130), (130, 330), (64, 129)])


def test_build_tiny_and_degenerate_trees(dev):
    trees = [make_tree(s) for s in ["(a,b)", "((a,b),c)", "(d,(e,(f,(g,(h,(a,b))))))", "(a,b,c,d)", "((a,b,c)x,(d,e))"]]
    names = sorted(so._all_tips(trees))
    tables = fl.flatten_trees(trees, [1, 2, 0.5, 1, 3], "depth", names)
    _build_and_compare(dev, tables)


@pytest.mark.parametrize(("name", "src", "exp", "weighting"), FILE_CASES, ids=[c[0] for c in FILE_CASES])
def test_build_reference_fixtures(dev, name, src, exp, weighting):
    trees = [make_tree(x.strip()) for x in (DATA_DIR / src).read_text().splitlines() if x.strip()]
    names = sorted(so._all_tips(trees))
    tables = fl.flatten_trees(trees, [1.0] * len(trees), weighting, names)
    _build_and_compare(dev, tables)


def test_build_batched_matches_single_batch(dev, monkeypatch):
    # force several tree batches through a tiny workspace limit: same bits
    tables = synthetic.make_tables(9, 500, 40, "branch", leaves_per_tree=400)
    monkeypatch.setenv("SCS_WS_LIMIT_MB", "1")
    small = Device(0)
    try:
        dtab = small.upload(tables)
        g = dtab.build()
        assert g.build_stats["n_batches"] > 1
        w = g.download()
        g.free()
        dtab.free()
    finally:
        small.close()
    w_ref, _ = to.pcg_dense(tables)
    assert np.array_equal(w, w_ref)


def test_build_config2_shape_properties(dev):
    # BASELINE.json config 2 (1000 taxa / 100 trees / depth): full size, checked
    # against the C oracle and through size-independent properties
    tables = synthetic.make_tables(0, 1000, 100, "depth")
    dtab = dev.upload(tables)
    g = dtab.build()
    w = g.download()
    deg = g.degrees()
    g.free()
    dtab.free()
    assert np.array_equal(w, w.T)
    assert np.all(np.diag(w) == 0)
    assert np.all(w == np.floor(w))  # depth weighting with unit tree weights is integer valued
    w_ref, _ = to.pcg_dense(tables)
    assert np.array_equal(w, w_ref)
    assert np.allclose(deg, w_ref.sum(axis=0), rtol=1e-13, atol=0)


def test_contract_matches_oracle(dev):
    trees = [make_tree(s) for s in ["(((a,b),(c,d)),(e,f))", "((a,b),((c,d),g))", "(((a,b),e),(c,d))", "((e,f),(a,(b,g)))"]]
    names = sorted(so._all_tips(trees))
    tables = fl.flatten_trees(trees, [1.0, 2.0, 1.0, 0.5], "branch", names)
    groups = fl.contraction_groups(tables)
    work, perm, group_start = relabel_for_contraction(tables, groups)
    w_ref, _ = to.pcg_dense(work)
    dtab = dev.upload(work)
    g = dtab.build().contract(group_start)
    got = g.download()
    g.free()
    dtab.free()
    assert np.array_equal(got, to.contract_dense(w_ref, group_start))


# ---------------------------------------------------------------------------
# operator and eigen-solve
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("b", [4, 8, 12, 16])
def test_symm_apply_matches_numpy(dev, b):
    tables = synthetic.make_tables(21, 515, 20, "branch", leaves_per_tree=480)
    w_ref, _ = to.pcg_dense(tables)
    s_ref, _ = to.normalized_operator(w_ref)
    dtab = dev.upload(tables)
    g = dtab.build()
    x = np.random.RandomState(b).standard_normal((515, b))
    y = g.apply(x)
    g.free()
    dtab.free()
    assert np.max(np.abs(y - s_ref @ x)) <= 1e-12


def _fiedler_case(dev, tables, block=0):
    w_ref, _ = to.pcg_dense(tables)
    ref = to.sign_flip_columns(so.spectral_maps(w_ref, np.random.RandomState(0)))
    dtab = dev.upload(tables)
    g = dtab.build()
    v0 = np.random.RandomState(0).uniform(-1, 1, tables.n_taxa)
    maps, stats = g.fiedler(v0, block=block)
    g.free()
    dtab.free()
    s_ref, dd = to.normalized_operator(w_ref)
    err_maps = float(np.max(np.abs(maps[:, 1] - ref[:, 1])))
    err_unit = float(np.max(np.abs(maps[:, 1] * dd - ref[:, 1] * dd)))
    err_col0 = float(np.max(np.abs(maps[:, 0] - ref[:, 0])))
    x = maps[:, 1] * dd
    x /= np.linalg.norm(x)
    true_res = float(np.linalg.norm(s_ref @ x - (x @ s_ref @ x) * x))
    info = dict(stats, err_maps=err_maps, err_unit=err_unit, err_col0=err_col0, true_res=true_res)
    print("FIEDLER", tables.n_taxa, tables.n_trees, info)
    assert err_maps <= FIEDLER_TOL, info
    assert err_unit <= FIEDLER_TOL, info  # the unit-norm eigenvector ARPACK returns: the stricter scale
    assert err_col0 <= FIEDLER_TOL, info
    assert true_res <= 1e-11, info
    return maps, stats, w_ref


@pytest.mark.parametrize(("n", "m", "strategy"), [(70, 10, "depth"), (200, 16, "branch"), (515, 30, "bootstrap"), (1000, 100, "depth")])
def test_fiedler_matches_sklearn(dev, n, m, strategy):
    tables = synthetic.make_tables(n, n, m, strategy)
    _fiedler_case(dev, tables)


@pytest.mark.parametrize("block", [4, 12, 16])
def test_fiedler_block_widths(dev, block):
    tables = synthetic.make_tables(3, 300, 24, "branch")
    _fiedler_case(dev, tables, block=block)


@pytest.mark.parametrize(("n", "m", "strategy", "k"), [(130, 150, "branch", None), (200, 300, "depth", 150),
                                                       (450, 260, "one", None), (700, 600, "branch", 500),
                                                       (300, 200, "bootstrap", 220), (140, 330, "bootstrap", None)])
def test_tree_parallel_build_of_a_small_node_is_the_walk_bit_for_bit(dev, n, m, strategy, k, monkeypatch):
    # a workgroup per (tile, tree) and the trees' cells added up in order afterwards
    # (k_sum_tree_tiles) against one workgroup per tile walking the trees: the same additions in
    # the same order -- and both the oracle's W.  700 x 600 needs two batches of cells (1 GiB each).
    tables = synthetic.make_tables(n + m, n, m, strategy, leaves_per_tree=k, random_weights=True)
    w_ref, _ = to.pcg_dense(tables)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SCS_TREE_PARALLEL", mode)
        dtab = dev.upload(tables)
        g = dtab.build()
        assert (g.build_stats["tree_parallel_batches"] > 0) == (mode == "1"), g.build_stats
        if n == 700 and mode == "1":
            assert g.build_stats["n_batches"] >= 2
        got[mode] = g.download()
        g.free()
        dtab.free()
    assert np.array_equal(got["0"], got["1"])
    assert np.array_equal(got["1"], w_ref)


def test_mid_size_nodes_with_many_trees_are_built_tree_parallel_by_default(dev, monkeypatch):
    monkeypatch.delenv("SCS_TREE_PARALLEL", raising=False)
    for n, m, expect, strategy in ((300, 200, True, "branch"), (300, 60, False, "branch"), (1500, 200, False, "branch"),
                                   (260, 150, True, "bootstrap")):
        tables = synthetic.make_tables(5, n, m, strategy)
        dtab = dev.upload(tables)
        g = dtab.build()
        assert (g.build_stats["tree_parallel_batches"] > 0) == expect, (n, m, g.build_stats)
        assert np.array_equal(g.download(), to.pcg_dense(tables)[0])
        g.free()
        dtab.free()


@pytest.mark.parametrize(("n", "block"), [(600, 4), (5000, 4), (700, 8), (4500, 8)])
def test_small_solves_in_front_of_the_tall_kernels_change_no_bit(dev, n, block, monkeypatch):
    # the LOBPCG loop with its three small solves inside the tall kernels (the default) against
    # the same loop with one-workgroup kernels of their own (SCS_SPLIT_SMALL=1): same partial
    # sums, same solves -- the same iterates, bit for bit
    tables = synthetic.make_tables(n + 1, n, 12, "branch", random_weights=True)
    dtab = dev.upload(tables)
    g = dtab.build()
    monkeypatch.delenv("SCS_SPLIT_SMALL", raising=False)
    monkeypatch.setenv("SCS_FOLD_PASS2", "0")  # (round 5's default folds pass 2 into the Gram kernel: below)
    maps_a, stats_a = g.fiedler(None, block=block)
    monkeypatch.setenv("SCS_SPLIT_SMALL", "1")
    maps_b, stats_b = g.fiedler(None, block=block)
    # round 5: the second orthonormalisation pass of R applied behind the SYMM stream, inside the Gram
    # kernel (to R and S R alike: it is linear) -- another sequence of roundings, the same eigenvector
    monkeypatch.delenv("SCS_SPLIT_SMALL")
    monkeypatch.delenv("SCS_FOLD_PASS2")
    maps_c, stats_c = g.fiedler(None, block=block)
    # ... and the first pass inside the SYMM launch (symmetric schedule, n >= 4096: the operator is then
    # applied to the raw residual block and S R comes out of both passes' transforms); off: the loop above
    monkeypatch.setenv("SCS_OVERLAP_PASS1", "0")
    maps_d, stats_d = g.fiedler(None, block=block)
    g.free()
    dtab.free()
    assert stats_a["iterations"] == stats_b["iterations"] > 3
    assert np.array_equal(maps_a, maps_b)
    scale = float(np.max(np.abs(maps_a[:, 1])))
    for maps_x, stats_x in ((maps_c, stats_c), (maps_d, stats_d)):
        assert stats_x["converged"] == 1 and abs(stats_x["iterations"] - stats_a["iterations"]) <= 3
        assert abs(stats_x["lambda"][1] - stats_a["lambda"][1]) <= 1e-13
        assert float(np.max(np.abs(maps_x[:, 1] - maps_a[:, 1]))) <= 1e-10 * scale
    if n < 4096:
        assert np.array_equal(maps_c, maps_d)  # no symmetric schedule, nothing to ride on


@pytest.mark.parametrize(("n", "m", "weights", "planted"), [(4096, 12, True, False), (5200, 40, False, False),
                                                               (6000, 6, True, True), (8192, 25, True, False)])
def test_mixed_precision_loop_reaches_the_same_pair(dev, monkeypatch, n, m, weights, planted):
    # round 5: at 4 096 <= V < 16 384 the loop's SYMM streams a single-precision image of W (search
    # directions only; S X / S P renewed through W, the confirmation through W) -- the same eigenpair to the
    # same tolerance as the all-double loop (SCS_LOWP=0) and as the mode that leaves the image after the
    # renewal (SCS_LOWP=1); the residual both report is the one measured through W itself
    tables = synthetic.make_tables(n + 7, n, m, "branch", random_weights=weights,
                                   planted_spr=int(np.ceil(0.02 * n)) if planted else None)
    dtab = dev.upload(tables)
    g = dtab.build()
    out = {}
    for mode in ("0", "2", "1"):
        monkeypatch.setenv("SCS_LOWP", mode)
        out[mode] = g.fiedler(None, block=4)
    # (the same graph again: the image is already there, the degrees are not computed a second time)
    monkeypatch.setenv("SCS_LOWP", "2")
    again = g.fiedler(None, block=4)
    g.free()
    dtab.free()
    maps0, st0 = out["0"]
    assert st0["n_apply32"] == 0 and st0["converged"] == 1
    scale = float(np.max(np.abs(maps0[:, 1])))
    for mode in ("2", "1"):
        maps, st = out[mode]
        assert st["converged"] == 1 and st["n_apply32"] > 0 and st["lowp_renewals"] <= 2, st
        assert abs(st["lambda"][1] - st0["lambda"][1]) <= 1e-13
        assert float(np.max(np.abs(maps[:, 1] - maps0[:, 1]))) <= 1e-10 * scale  # (the bar itself; vs scikit-learn: test_gpu_mixed_precision.py)
        assert st["iterations"] <= st0["iterations"] + 6, (st, st0)
    assert out["2"][1]["n_apply32"] >= out["1"][1]["n_apply32"]
    assert np.array_equal(again[0], out["2"][0])


@pytest.mark.parametrize(("seed", "n", "m", "strategy"), [(11, 4501, 3, "branch"), (12, 6564, 3, "depth"),
                                                          (13, 5446, 6, "depth")])
def test_symmetric_schedule_with_isolated_vertices_converges(dev, monkeypatch, seed, n, m, strategy):
    # partial coverage by a few trees: taxa of degree 0, both leading pairs iterated, and the trivial one
    # converges to rounding level long before the other -- the loop with pass 1 inside the SYMM launch then
    # composes S R from terms far larger than R itself; with the mean of both Gram entries it let that
    # column's noise grow until the block blew up (round 5, found by tools/fuzz_mixed_precision.py: 2 000
    # iterations, lambda 1e145).  All-double and image loop alike: a few dozen iterations, the same pair.
    tables = synthetic.make_tables(seed, n, m, strategy, leaves_per_tree=int(0.6 * n))
    dtab = dev.upload(tables)
    g = dtab.build()
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("SCS_LOWP", mode)
        out[mode] = g.fiedler(None)
    g.free()
    dtab.free()
    (m0, s0), (m2, s2) = out["0"], out["2"]
    assert s0["used_constraint"] == 0
    for st in (s0, s2):
        assert st["converged"] == 1 and st["iterations"] < 150, st
    assert abs(s0["lambda"][1] - s2["lambda"][1]) <= 1e-13
    if abs(s0["lambda"][1] - s0["lambda_next"]) > 1e-6:
        assert float(np.max(np.abs(m0[:, 1] - m2[:, 1]))) <= 1e-10 * float(np.max(np.abs(m0[:, 1])))


@pytest.mark.parametrize("n", [3, 4, 8, 33, 64, 65, 80, 96])
def test_fiedler_small_dense_path(dev, n):
    # generic (random tree weights, branch lengths) so that no eigenvalue is repeated
    tables = synthetic.make_tables(n, n, 9, "branch", random_weights=True)
    w_ref, _ = to.pcg_dense(tables)
    lam = np.sort(np.linalg.eigvalsh(to.normalized_operator(w_ref)[0]))[::-1]
    assert lam[1] - lam[2] > 1e-6 if n > 2 else True
    ref = to.sign_flip_columns(so.spectral_maps(w_ref, np.random.RandomState(0)))
    dtab = dev.upload(tables)
    g = dtab.build()
    maps, stats = g.fiedler(None)
    g.free()
    dtab.free()
    assert stats["block"] == 0
    assert np.max(np.abs(maps - ref)) <= FIEDLER_TOL, (maps, ref)


@pytest.mark.parametrize("n", [40, 90, 150])
def test_fiedler_with_isolated_vertices(dev, n):
    # taxa that never share a root side with anything have degree 0: scipy scales their rows by 1
    # (reference: scipy/sparse/csgraph/_laplacian.py:550-557), the trivial eigenvector is no
    # longer sqrt(d) of ALL vertices, and the solver iterates both leading pairs itself
    import random

    from tests.test_treearrays import random_tree

    rng = random.Random(n)
    names = [f"t{i:03d}" for i in range(n)]
    loners, rest = names[:1], names[1:]  # one loner: two would make the second pair degenerate
    trees = []
    for _ in range(12):
        inner = random_tree(rng, rest, multifurcate=0.0, none_len=0.0, none_sup=0.0, unary=0.0)
        extra = random_tree(rng, rng.sample(rest, 5), multifurcate=0.0, none_len=0.0, none_sup=0.0, unary=0.0)
        trees.append(fl_tree([fl_leaf(x) for x in loners] + [inner]) if _ % 2 else fl_tree([fl_leaf(loners[0]), extra]))
    tables = fl.flatten_trees(trees, [1.0 + 0.1 * i for i in range(12)], "branch", names)
    w_ref, _ = to.pcg_dense(tables)
    assert np.count_nonzero(w_ref.sum(axis=1) == 0) == 1
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = to.sign_flip_columns(so.spectral_maps(w_ref, np.random.RandomState(0)))
    dtab = dev.upload(tables)
    g = dtab.build()
    assert np.array_equal(g.download(), w_ref)
    maps, stats = g.fiedler(np.random.RandomState(0).uniform(-1, 1, n))
    g.free()
    dtab.free()
    if n > 64:
        assert stats["used_constraint"] == 0 and stats["converged"] == 1, stats
    assert np.max(np.abs(maps - ref)) <= FIEDLER_TOL, stats


def test_labels_match_spectral_clustering(dev):
    from spectralclustersupertree_amd.scs import spectral_bipartition_device

    for seed, (n, m, strategy) in enumerate([(120, 12, "depth"), (400, 20, "branch")]):
        tables = synthetic.make_tables(40 + seed, n, m, strategy)
        w_ref, _ = to.pcg_dense(tables)
        want = so.spectral_labels(w_ref, np.random.RandomState(seed))
        members, got = spectral_bipartition_device(
            tables, np.random.RandomState(seed), contract_edges=False, device=dev
        )
        assert [int(mm[0]) for mm in members] == list(range(n))
        assert np.array_equal(got, want), f"{int(np.sum(got != want))} labels differ"


def test_two_vertices_after_contraction(dev):
    # V == 2 after contraction (SURVEY.md 8a edge cases): scipy's eigsh falls back to a dense
    # eigh with a RuntimeWarning and the two vertices end up in different parts
    import warnings

    from spectralclustersupertree_amd import construct_supertree
    from spectralclustersupertree_amd.scs import spectral_bipartition_device

    trees = [make_tree(s) for s in ["(((a,b),(c,d)));", "((a,b),(c,d));"]]
    names = sorted(so._all_tips(trees))
    tables = fl.flatten_trees(trees, [1.0, 1.0], "branch", names)
    assert int(fl.pcg_components(tables).max()) == 0
    groups = fl.contraction_groups(tables)
    assert int(groups.max()) + 1 == 2
    report = {}
    members, labels = spectral_bipartition_device(tables, np.random.RandomState(3), contract_edges=True,
                                                  device=dev, report=report)
    assert sorted(sorted(int(i) for i in mm) for mm in members) == [[0, 1], [2, 3]]
    assert labels[0] != labels[1]
    assert report["n_vertices"] == 2 and report["block"] == 0
    # the embedding itself against scikit-learn on the contracted 2 x 2 matrix
    work, perm, group_start = relabel_for_contraction(tables, groups)
    dtab = dev.upload(work)
    g = dtab.build().contract(group_start)
    w2 = g.download()
    maps, _ = g.fiedler(None)
    g.free()
    dtab.free()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = to.sign_flip_columns(so.spectral_maps(w2, np.random.RandomState(0)))
    assert np.max(np.abs(maps - ref)) <= FIEDLER_TOL, (maps, ref)
    for seed in range(3):
        got = construct_supertree(trees, pcg_weighting="branch", random_state=np.random.RandomState(seed))
        assert got.sorted().same_shape(make_tree("((a,b),(c,d));").sorted())


def test_two_squares_repeated_eigenvalue_partition():
    # lambda2 is repeated for this input (reference: tests/test_spectral_cluster_supertree.py:80-106):
    # the Fiedler vector is not unique, only the partition is -- every seed must give the
    # reference's topology and the solver must report convergence
    from spectralclustersupertree_amd import construct_supertree

    case = next(c for c in INLINE_CASES if c.name == "two_squares")
    trees = [make_tree(s) for s in case.trees]
    expected = make_tree(case.expected)
    for seed in range(8):
        got = construct_supertree(trees, weights=case.weights, pcg_weighting=case.pcg_weighting,
                                  contract_edges=case.contract_edges,
                                  random_state=np.random.RandomState(seed))
        assert got.sorted().same_shape(expected.sorted()), seed


def test_unconverged_solve_is_reported_not_clustered(dev):
    # scs_fiedler stops above tol -> SCS_ENOCONV with the block it reached; the recursion's
    # wrapper retries wider, then refuses a residual that is still far off
    import warnings

    from spectralclustersupertree_amd._native import ConvergenceError
    from spectralclustersupertree_amd.scs import _fiedler_checked, spectral_bipartition_device

    tables = synthetic.make_tables(12, 600, 20, "branch")
    dtab = dev.upload(tables)
    g = dtab.build()
    v0 = np.random.RandomState(0).uniform(-1, 1, 600)
    with pytest.raises(ConvergenceError) as ei:
        g.fiedler(v0, max_iter=2)
    assert ei.value.code == nv.ENOCONV and ei.value.stats["converged"] == 0
    assert ei.value.maps.shape == (600, 2) and ei.value.stats["resid"][1] > 1e-13
    # a target below the floating-point floor: the solver stagnates around 1e-14, the wrapper
    # retries, then accepts the block with a warning
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        maps, stats = _fiedler_checked(g, v0, 1e-30, 300, 0)
    assert any(issubclass(w.category, RuntimeWarning) for w in caught)
    assert stats["converged"] == 0 and max(stats["resid"]) <= 1e-11
    maps_ok, _ = g.fiedler(v0)
    assert np.max(np.abs(maps - maps_ok)) <= 1e-9
    g.free()
    dtab.free()
    # far from converged even after the retry (1 and then 4 iterations): refused
    with pytest.raises(RuntimeError, match="did not converge"):
        spectral_bipartition_device(tables, np.random.RandomState(0), contract_edges=False, device=dev,
                                    max_iter=1)


def test_contract_more_groups_than_a_grid_dimension(dev):
    # > 65 535 contracted rows: the row index of k_contract is a grid-stride loop (a grid.y of
    # that size is not launchable); groups of one or two taxa, checked against numpy on samples
    n = 70000
    tables = synthetic.make_tables(4, n, 2, "depth", leaves_per_tree=300)
    gs = [0]
    rs = np.random.RandomState(1)
    while gs[-1] < n:
        gs.append(min(n, gs[-1] + (2 if rs.rand() < 0.05 else 1)))
    group_start = np.asarray(gs, dtype=np.int32)
    n_groups = len(group_start) - 1
    assert n_groups > 65536
    dtab = dev.upload(tables)
    g = dtab.build()
    rows_old = {int(r): g.download_rows(int(r), 1)[0] for r in range(0, 8)}
    first_pair = int(np.argmax(np.diff(group_start) == 2))
    for r in (group_start[first_pair], group_start[first_pair] + 1, group_start[-2], n - 1):
        rows_old[int(r)] = g.download_rows(int(r), 1)[0]
    c = g.contract(group_start)
    assert c.shape == (n_groups, 0, n_groups)
    for grp in (0, first_pair, n_groups - 1, 65535, 65536, 66000):
        got = c.download_rows(grp, 1)[0]
        lo, hi = int(group_start[grp]), int(group_start[grp + 1])
        if not all(r in rows_old for r in range(lo, hi)):
            continue
        member_rows = np.vstack([rows_old[r] for r in range(lo, hi)])
        want = np.maximum.reduceat(member_rows.max(axis=0), group_start[:-1])
        want[grp] = 0.0
        assert np.array_equal(got, want), grp
    c.free()
    dtab.free()


def _run_local_group(tables, splits, shared, v0):
    """Row-partitioned build + solve on ONE GPU: `world` contexts, one host thread each,
    the in-process communicator instead of RCCL.  Returns [(W rows, maps, stats, build stats)]."""
    world = len(splits) - 1
    lib = nv.load_library()
    group = nv.C.c_void_p()
    nv.check(lib.scs_local_group_create(world, nv.C.byref(group)))
    out = [None] * world
    err = [None] * world

    def worker(rank):
        try:
            d = Device(0, rank, world, _local_group=group)
            dtab = d.upload(tables)
            g = dtab.build(splits[rank], splits[rank + 1], shared=shared)
            w = g.download()
            maps, stats = g.fiedler(v0)
            out[rank] = (w, maps, stats, g.build_stats)
            g.free()
            dtab.free()
            d.close()
        except BaseException as e:  # noqa: BLE001
            err[rank] = e

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    lib.scs_local_group_destroy(group)
    assert err == [None] * world, err
    return out


@pytest.mark.parametrize(
    "splits,shared,strategy",
    [
        ([0, 333, 700], False, "branch"),
        ([0, 333, 700], True, "branch"),
        ([0, 333, 700], True, "bootstrap"),  # general kernel writes the packed tiles
        ([0, 100, 290, 700], True, "branch"),  # three ranks, uneven rows, a spare tile slot
        ([0, 64, 700], True, "depth"),
    ],
)
def test_multi_rank_row_partition_matches_single(dev, splits, shared, strategy):
    tables = synthetic.make_tables(77, 700, 24, strategy, leaves_per_tree=650)
    w_ref, _ = to.pcg_dense(tables)
    ref = to.sign_flip_columns(so.spectral_maps(w_ref, np.random.RandomState(0)))
    v0 = np.random.RandomState(0).uniform(-1, 1, 700)
    out = _run_local_group(tables, splits, shared, v0)
    for rank, (w, maps, stats, bstats) in enumerate(out):
        assert bstats["symmetric"] == (2 if shared else 0)
        assert np.array_equal(w, w_ref[splits[rank] : splits[rank + 1]]), rank
        assert np.max(np.abs(maps[:, 1] - ref[:, 1])) <= FIEDLER_TOL, stats
        assert np.array_equal(maps, out[0][1])  # every rank holds the same embedding
    if shared:
        # every cell evaluated once across the job: the ranks' tile counts add up to the
        # single-GPU symmetric schedule's
        single = dev.upload(tables).build()
        assert sum(o[3]["n_tiles"] for o in out) == single.build_stats["n_tiles"]
        single.free()


@pytest.mark.parametrize("mode", ["p2p", "allgather"])
def test_shared_build_exchange_modes(dev, monkeypatch, mode):
    # the point-to-point exchange (a tile goes only to the ranks whose rows it touches) and the
    # all-gather of the whole packed triangle give the same rows; the former moves fewer bytes
    if mode == "allgather":
        monkeypatch.setenv("SCS_EXCHANGE", "allgather")
    tables = synthetic.make_tables(31, 1500, 8, "branch", leaves_per_tree=1400)
    w_ref, _ = to.pcg_dense(tables)
    splits = [0, 320, 700, 1100, 1500]
    out = _run_local_group(tables, splits, True, None)
    assert np.array_equal(np.vstack([o[0] for o in out]), w_ref)
    received = [o[3]["exchange_bytes"] for o in out]
    n_tiles_all = sum(o[3]["n_tiles"] for o in out)
    tile_bytes = out[0][3]["cell_trees"] / out[0][3]["n_tiles"] / tables.n_trees * 8  # cells per tile x 8
    if mode == "allgather":
        assert all(rb >= 0.7 * n_tiles_all * tile_bytes for rb in received)
    else:
        assert all(0 < rb < 0.75 * n_tiles_all * tile_bytes for rb in received), received


def test_shared_build_multi_batch(dev, monkeypatch):
    # a workspace small enough to force several tree batches: the packed tiles carry the
    # running sums between batches
    monkeypatch.setenv("SCS_WS_LIMIT_MB", "2")
    tables = synthetic.make_tables(5, 900, 40, "branch", leaves_per_tree=800)
    w_ref, _ = to.pcg_dense(tables)
    out = _run_local_group(tables, [0, 500, 900], True, None)
    assert out[0][3]["n_batches"] > 1
    assert np.array_equal(np.vstack([o[0] for o in out]), w_ref)


# ---------------------------------------------------------------------------
# the producer / consumer tile kernel (scs_mono_wide.h: k_accumulate_spec -- two column tiles of a
# row block per workgroup, eight consumer waves with the table expansion woven into the cell
# loop, four producer waves running the column step two trees ahead).  scs_pcg_build takes it by
# itself once the groups fill the chip (the configs[2]-size tests and the bench run it);
# SCS_WIDE=1 puts the small shapes through it, SCS_WIDE=0 forces the 4-wave kernel.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("strategy", ["one", "depth", "branch"])
@pytest.mark.parametrize(("n", "m", "k"), [(37, 6, 20), (300, 12, 300), (700, 7, 512), (1100, 9, 900),
                                           (1537, 5, 1537), (330, 300, 200), (640, 1, 640), (640, 2, 600),
                                           (900, 3, 700), (900, 4, 900), (900, 5, 650), (900, 6, 900)])
def test_spec_kernel_bit_exact(dev, monkeypatch, strategy, n, m, k):
    # one to seven column tiles per row block: full and half groups, ragged last tile, taxa
    # missing from trees, a single tree batch of 300 trees -- and one to six trees: every length of
    # the producers' and the records' pipelines (queries two trees ahead, records five)
    monkeypatch.setenv("SCS_WIDE", "1")
    tables = synthetic.make_tables(200 + n + m, n, m, strategy, leaves_per_tree=k, random_weights=(n % 2 == 0))
    assert tables.monotone
    _build_and_compare(dev, tables, expect_spec=True)


def test_spec_kernel_row_blocks_and_batches(dev, monkeypatch):
    # the non-symmetric schedule of row-partitioned ranks (no mirror image, every column tile of
    # a row block), and several tree batches through a tiny workspace (the sums travel through W)
    monkeypatch.setenv("SCS_WIDE", "1")
    tables = synthetic.make_tables(6, 1000, 10, "branch", leaves_per_tree=800, random_weights=True)
    _build_and_compare(dev, tables, [(0, 100), (100, 1000), (64, 65), (7, 601)], expect_spec=True)
    monkeypatch.setenv("SCS_WS_LIMIT_MB", "1")
    small = Device(0)
    try:
        tables = synthetic.make_tables(9, 900, 40, "branch", leaves_per_tree=700)
        dtab = small.upload(tables)
        g = dtab.build()
        assert g.build_stats["n_batches"] > 1
        assert g.build_stats["spec_batches"] == g.build_stats["n_batches"], g.build_stats
        w = g.download()
        g.free()
        dtab.free()
    finally:
        small.close()
    w_ref, _ = to.pcg_dense(tables)
    assert np.array_equal(w, w_ref)


def test_spec_kernel_shared_multi_rank_build(dev, monkeypatch):
    # packed tiles of the shared multi-rank build (a rank's tiles of a row block are every
    # world-th column tile), several batches, and row-partitioned ranks
    monkeypatch.setenv("SCS_WIDE", "1")
    monkeypatch.setenv("SCS_WS_LIMIT_MB", "2")
    tables = synthetic.make_tables(5, 1300, 30, "branch", leaves_per_tree=1200)
    w_ref, _ = to.pcg_dense(tables)
    out = _run_local_group(tables, [0, 500, 1300], True, None)
    assert out[0][3]["n_batches"] > 1
    assert all(o[3]["spec_batches"] == o[3]["n_batches"] for o in out), [o[3] for o in out]
    assert np.array_equal(np.vstack([o[0] for o in out]), w_ref)
    out = _run_local_group(tables, [0, 333, 800, 1300], False, None)
    assert all(o[3]["spec_batches"] == o[3]["n_batches"] for o in out), [o[3] for o in out]
    assert np.array_equal(np.vstack([o[0] for o in out]), w_ref)


def test_spec_and_four_wave_kernels_agree_at_size(dev, monkeypatch):
    # 3 000 taxa / 600 trees: 12 column tiles per row block, several tree batches by the tree cap
    # (256 for the 4-wave kernel, 512 -- equal lengths -- for the producer / consumer kernel);
    # the two kernels against each other, whole matrix, and sampled rows against the oracle
    tables = synthetic.make_tables(3, 3000, 600, "branch", random_weights=True)
    got = {}
    for wide in ("0", "1", None):
        if wide is None:
            monkeypatch.delenv("SCS_WIDE")
        else:
            monkeypatch.setenv("SCS_WIDE", wide)
        dtab = dev.upload(tables)
        g = dtab.build()
        got[wide] = g.download()
        assert g.build_stats["n_batches"] > 1
        # (left to itself scs_pcg_build takes the producer / consumer kernel from 25 tiles on since
        # round 4: a walk of a few hundred tiles is bound by the latency of a tree's step)
        assert g.build_stats["spec_batches"] == (0 if wide == "0" else g.build_stats["n_batches"])
        g.free()
        dtab.free()
    assert np.array_equal(got["0"], got["1"]) and np.array_equal(got["0"], got[None])
    rows = np.unique(np.random.RandomState(2).randint(0, 3000, size=12)).astype(np.int32)
    assert np.array_equal(got["1"][rows], to.pcg_rows(tables, rows))
    assert np.array_equal(got["1"], got["1"].T) and not np.any(np.diag(got["1"]))


def test_small_node_that_fails_is_solved_again_on_the_general_path(dev, monkeypatch):
    # a node of 65 .. 128 taxa whose batched one-sided Jacobi reports NaN eigenvalues (forced here:
    # backend._FAIL_SMALL_FOR_TESTS) goes through upload + build + LOBPCG instead of failing the whole batch;
    # its neighbours in the batch keep their batched results
    nodes = [(synthetic.make_tables(70 + i, n, 12, "branch"), None) for i, n in enumerate((40, 90, 120))]
    want = dev.small_solve(nodes, want_w=True)
    from spectralclustersupertree_amd import backend

    monkeypatch.setattr(backend, "_FAIL_SMALL_FOR_TESTS", True)
    got = dev.small_solve(nodes, want_w=True)
    assert np.array_equal(got[0][0], want[0][0]) and np.array_equal(got[0][1], want[0][1])
    for g, w in zip(got[1:], want[1:]):
        assert np.array_equal(g[2], w[2])  # the same W
        assert np.max(np.abs(g[1][:2] - w[1][:2])) <= 1e-12  # the same eigenvalues
        assert np.max(np.abs(g[0][:, 1] - w[0][:, 1])) <= 1e-10 * np.max(np.abs(w[0][:, 1]))


@pytest.mark.parametrize("strategy", ["one", "depth", "branch"])
def test_partial_coverage_forests_walk_per_tile_tree_lists(dev, monkeypatch, strategy):
    # trees that hold well under 1 / 64 of the taxa: every tile walks only the trees that touch it
    # (k_tile_lists + k_accumulate_mono<.., .., true>) -- the skipped steps would add +0.0: same bits.
    # Taxa no tree holds, tiles no tree touches, several tree batches (the sums travel through W),
    # row ranges of a row-partitioned rank; then the same through the switch on a denser forest.
    monkeypatch.delenv("SCS_TILE_LISTS", raising=False)
    tables = synthetic.make_tables(41, 3000, 300, strategy, leaves_per_tree=20, random_weights=True)
    assert tables.monotone
    w_ref, _ = to.pcg_dense(tables)
    dtab = dev.upload(tables)
    g = dtab.build()
    assert g.build_stats["listed_batches"] == g.build_stats["n_batches"] >= 1 and g.build_stats["spec_batches"] == 0
    assert np.array_equal(g.download(), w_ref)
    g.free()
    for rb, re_ in ((0, 1000), (1024, 3000), (64, 65)):
        g = dtab.build(rb, re_)
        assert g.build_stats["listed_batches"] >= 1
        assert np.array_equal(g.download(), w_ref[rb:re_])
        g.free()
    dtab.free()
    monkeypatch.setenv("SCS_WS_LIMIT_MB", "1")
    small = Device(0)
    try:
        dtab = small.upload(tables)
        g = dtab.build()
        assert g.build_stats["n_batches"] > 1 and g.build_stats["listed_batches"] == g.build_stats["n_batches"]
        assert np.array_equal(g.download(), w_ref)
        g.free()
        dtab.free()
    finally:
        small.close()
    monkeypatch.delenv("SCS_WS_LIMIT_MB")
    monkeypatch.setenv("SCS_TILE_LISTS", "1")
    tables = synthetic.make_tables(42, 900, 40, strategy, leaves_per_tree=300)
    dtab = dev.upload(tables)
    g = dtab.build()
    assert g.build_stats["listed_batches"] >= 1
    assert np.array_equal(g.download(), to.pcg_dense(tables)[0])
    g.free()
    dtab.free()


def test_rccl_world_of_one(dev):
    # exercises the RCCL binding (dlopen, unique id, comm init, all-gather) on one GPU
    uid = Device.unique_id()
    d = Device(0, 0, 1, uid)
    try:
        tables = synthetic.make_tables(8, 200, 10, "depth")
        dtab = d.upload(tables)
        g = dtab.build()
        maps, stats = g.fiedler(None)
        g.free()
        dtab.free()
        # the point-to-point wrappers of the shared build's tile exchange (ncclGroupStart,
        # ncclSend, ncclRecv, ncclGroupEnd) and ncclAllGather itself, on this one rank
        x = np.random.RandomState(3).standard_normal(100_003)
        back = d.comm_selftest(x)
    finally:
        d.close()
    assert stats["converged"] == 1
    assert np.array_equal(back, x)


# ---------------------------------------------------------------------------
# end to end: the reference's own test cases through construct_supertree
# ---------------------------------------------------------------------------
def _scs(trees, expected, **kw):
    from spectralclustersupertree_amd import construct_supertree

    for seed in (0, 1):
        got = construct_supertree(trees, random_state=np.random.RandomState(seed), **kw)
        assert got.sorted().same_shape(expected.sorted()), f"{got} != {expected}"


@pytest.mark.parametrize("case", INLINE_CASES, ids=lambda c: c.name)
def test_reference_inline_cases_end_to_end(case):
    _scs(
        [make_tree(s) for s in case.trees],
        make_tree(case.expected),
        weights=case.weights,
        pcg_weighting=case.pcg_weighting,
        contract_edges=case.contract_edges,
    )


@pytest.mark.parametrize(("name", "src", "exp", "weighting"), FILE_CASES, ids=[c[0] for c in FILE_CASES])
def test_reference_fixtures_end_to_end(name, src, exp, weighting):
    from spectralclustersupertree_amd import load_trees

    _scs(load_trees(DATA_DIR / src), load_tree(DATA_DIR / exp), pcg_weighting=weighting)


@pytest.mark.parametrize(("name", "src", "exp", "weighting"), FILE_CASES, ids=[c[0] for c in FILE_CASES])
def test_reference_fixtures_end_to_end_from_tree_arrays(name, src, exp, weighting):
    # the same fixtures through the C Newick loader: no tree object is built for the input
    from spectralclustersupertree_amd.load import load_tree_arrays

    _scs(load_tree_arrays(DATA_DIR / src), load_tree(DATA_DIR / exp), pcg_weighting=weighting)


def test_reference_not_completed_end_to_end():
    case = NOT_COMPLETED_CASE
    trees = [make_tree(s) for s in case.trees] + [NotCompleted("ERROR", "local", "Example NotCompleted")]
    _scs(trees, make_tree(case.expected), weights=case.weights)


# ---------------------------------------------------------------------------
# batched small nodes
# ---------------------------------------------------------------------------
def _supertriplets_spectral_nodes():
    """Every node of the reference's supertriplets fixture at which the reference calls
    SpectralClustering (47 of them, V = 100 ... 3), as (tables, group_start) in the
    recursion's own numbering, by walking the recursion with the oracle's bipartition."""
    from tests.test_treearrays import cpu_bipartition

    from spectralclustersupertree_amd import scs
    from spectralclustersupertree_amd.load import load_tree_arrays

    name, src, exp, weighting = next(c for c in FILE_CASES if "supertriplets" in c[0])
    nodes = []

    def spy(tables, random_state, *, contract_edges):
        groups = fl.contraction_groups(tables) if contract_edges else np.arange(tables.n_taxa, dtype=np.int32)
        if int(groups.max()) + 1 < tables.n_taxa:
            work, _, group_start = relabel_for_contraction(tables, groups)
        else:
            work, group_start = tables, None
        nodes.append((work, group_start))
        return cpu_bipartition(tables, random_state, contract_edges=contract_edges)

    scs._construct(load_tree_arrays(DATA_DIR / src), weighting, True, np.random.RandomState(0), spy)
    return nodes


def test_small_solve_batch_matches_sklearn_on_the_supertriplets_nodes(dev):
    import warnings

    nodes = _supertriplets_spectral_nodes()
    assert len(nodes) == 47
    small = [nd for nd in nodes if nd[0].n_taxa <= dev.SMALL_MAX_TAXA]
    assert len(small) >= 40
    out = dev.small_solve(small, want_w=True)
    worst = 0.0
    for (tables, gs), (maps, lam, w) in zip(small, out):
        w_ref, _ = to.pcg_dense(tables)
        if gs is not None:
            w_ref = to.contract_dense(w_ref, gs)
        assert np.array_equal(w, w_ref)  # W bit-exact, contraction included
        s_ref, _ = to.normalized_operator(w_ref)
        ev = np.sort(np.linalg.eigvalsh(s_ref))[::-1]
        assert np.max(np.abs(lam[: min(3, len(ev))] - ev[:3])) <= 1e-12
        if len(ev) > 2 and ev[1] - ev[2] < 1e-6:
            continue  # repeated lambda2: only the partition is defined, covered end to end
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = to.sign_flip_columns(so.spectral_maps(w_ref, np.random.RandomState(0)))
        # the sign rule is ill-defined when the two largest |entries| tie: accept either sign
        for c in range(2):
            err = float(np.max(np.abs(maps[:, c] - ref[:, c])))
            mags = np.sort(np.abs(ref[:, c]))[::-1]
            if mags[0] - mags[1] <= 1e-9 * mags[0]:
                err = min(err, float(np.max(np.abs(maps[:, c] + ref[:, c]))))
            worst = max(worst, err)
            assert err <= FIEDLER_TOL, (tables.n_taxa, c, err)
    print("SMALL BATCH nodes", len(small), "worst |maps - sklearn|", worst)


def test_small_solve_equals_the_per_node_path(dev):
    # the fused kernel against build + contract + fiedler on the same nodes, one at a time
    rs = np.random.RandomState(4)
    nodes = []
    for i in range(20):
        # (up to 64 taxa: two-sided Jacobi in LDS; 65 .. 128: the one-sided one, round 4)
        n = int(rs.randint(3, 65)) if i < 12 else int(rs.randint(65, 129))
        nodes.append((synthetic.make_tables(500 + i, n, int(rs.randint(2, 30)), ["one", "depth", "branch", "bootstrap"][i % 4],
                                            leaves_per_tree=max(2, n - int(rs.randint(0, 3))),
                                            random_weights=bool(i % 2)), None))
    out = dev.small_solve(nodes, want_w=True)
    for (tables, _), (maps, lam, w) in zip(nodes, out):
        dtab = dev.upload(tables)
        g = dtab.build()
        assert np.array_equal(g.download(), w)
        ref, stats = g.fiedler(None)
        g.free()
        dtab.free()
        deg = w.sum(axis=0)
        s_ref, _ = to.normalized_operator(w)
        ev = np.sort(np.linalg.eigvalsh(s_ref))[::-1]
        if np.all(deg > 0) and len(ev) > 2 and ev[1] - ev[2] > 1e-6 and ev[0] - ev[1] > 1e-6:
            assert np.max(np.abs(maps - ref)) <= FIEDLER_TOL, tables.n_taxa
        assert abs(lam[1] - stats["lambda"][1]) <= 1e-12


@pytest.mark.parametrize("n,m,k,strategy", [(5, 1500, 4, "branch"), (12, 700, 9, "bootstrap"), (40, 333, 25, "branch"),
                                            (64, 150, 64, "depth"), (33, 37, 20, "one"),
                                            # 65 .. 128 taxa (round 4): trees of up to 128 leaves, odd and even sizes
                                            (65, 90, 65, "branch"), (96, 120, 90, "branch"), (127, 40, 127, "bootstrap"),
                                            (128, 200, 128, "depth"), (100, 1500, 70, "branch"), (113, 17, 66, "one")])
def test_small_solve_many_trees_bit_exact(dev, n, m, k, strategy):
    # one node spread over many workgroups (runs of trees -> addends -> tree-ordered sums):
    # W bit for bit the oracle's, in a batch with a second node of another shape
    tables = synthetic.make_tables(900 + n, n, m, strategy, leaves_per_tree=k, random_weights=True)
    other = synthetic.make_tables(77, 9, 21, "branch", leaves_per_tree=7)
    out = dev.small_solve([(tables, None), (other, None), (tables, None)], want_w=True)
    for tb, (maps, lam, w) in zip((tables, other, tables), out):
        w_ref, _ = to.pcg_dense(tb)
        assert np.array_equal(w, w_ref)
        assert np.array_equal(w, w.T)
    assert np.array_equal(out[0][0], out[2][0])  # the same node twice: the same bits


def test_small_solve_nodes_of_65_to_128_taxa_match_sklearn(dev):
    # the batched path above the two-sided Jacobi's 64 vertices (V <= 128):
    # W bit for bit incl. contraction (planted twins), eigenvalues and embedding against
    # scikit-learn at 1e-10, several such nodes and small ones in one batch
    import warnings

    rs = np.random.RandomState(11)
    nodes = []
    for i, n in enumerate([65, 66, 80, 97, 111, 128, 30, 128]):
        m = int(rs.randint(8, 60))
        tables = synthetic.make_tables(1300 + i, n, m, ["branch", "depth", "bootstrap", "one"][i % 4],
                                       leaves_per_tree=max(2, n - int(rs.randint(0, 9))), random_weights=bool(i % 2))
        gs = None
        if i in (2, 5):  # contraction groups: some consecutive taxa merged
            cuts = np.sort(rs.choice(np.arange(1, n), size=n - 1 - 7, replace=False))
            gs = np.concatenate(([0], cuts, [n])).astype(np.int32)
        nodes.append((tables, gs))
    out = dev.small_solve(nodes, want_w=True)
    worst = 0.0
    for (tables, gs), (maps, lam, w) in zip(nodes, out):
        w_ref, _ = to.pcg_dense(tables)
        if gs is not None:
            w_ref = to.contract_dense(w_ref, gs)
        assert np.array_equal(w, w_ref), tables.n_taxa
        s_ref, _ = to.normalized_operator(w_ref)
        ev = np.sort(np.linalg.eigvalsh(s_ref))[::-1]
        assert np.max(np.abs(lam - ev[:3])) <= 1e-12, (tables.n_taxa, lam, ev[:3])
        if ev[1] - ev[2] < 1e-6 or ev[0] - ev[1] < 1e-6:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = to.sign_flip_columns(so.spectral_maps(w_ref, np.random.RandomState(0)))
        err = float(np.max(np.abs(maps - ref)))
        worst = max(worst, err)
        assert err <= FIEDLER_TOL, (tables.n_taxa, err)
    print("SMALL BATCH 65..128 taxa: worst |maps - sklearn|", worst)


def test_small_solve_begin_end_tickets(dev):
    # scs_small_solve_begin / _end: several solves outstanding on one context, ended in any order,
    # one dropped unasked-for -- the same bits as the waited-for call, slots reused
    from spectralclustersupertree_amd import _native as nv

    rs = np.random.RandomState(8)
    nodes = []
    for i in range(9):
        n = int(rs.randint(3, 65))
        nodes.append((synthetic.make_tables(700 + i, n, int(rs.randint(3, 400)), ["branch", "bootstrap", "one"][i % 3],
                                            leaves_per_tree=max(2, n - int(rs.randint(0, 3))),
                                            random_weights=bool(i % 2)), None))
    want = dev.small_solve(nodes, want_w=True)
    for round_ in range(3):
        tickets = [dev.small_solve_begin([nd], want_w=True) for nd in nodes]
        dropped = tickets.pop(4)
        del dropped  # never asked for: its slot goes back
        order = list(range(len(tickets)))
        rs.shuffle(order)
        for j in order:
            i = j if j < 4 else j + 1
            (maps, lam, w), = tickets[j].result()
            assert np.array_equal(maps, want[i][0]) and np.array_equal(lam, want[i][1])
            assert np.array_equal(w, want[i][2])
            assert tickets[j].result()[0][0] is maps  # kept, not fetched twice
    # a batch as one ticket next to single ones
    a = dev.small_solve_begin(nodes[:3])
    b = dev.small_solve_begin(nodes[3:4])
    assert np.array_equal(b.result()[0][0], want[3][0])
    for i, (maps, lam) in enumerate(a.result()):
        assert np.array_equal(maps, want[i][0])
    with pytest.raises(nv.ScsError, match="no such ticket"):
        nv.check(dev._lib.scs_small_solve_end(dev._ctx, 12345, None, None, None))


@pytest.mark.parametrize("n,m,k,strategy", [(1000, 100, 1000, "depth"), (700, 40, 600, "branch"), (130, 9, 100, "one")])
def test_atomic_scatter_variant_agrees_with_the_ordered_build(dev, n, m, k, strategy):
    # SURVEY.md section 7 (ii) / the north star's literal wording: input-stationary scatter with
    # fp64 atomicAdd, kept as the comparison variant; agrees to <= 1e-12 relative with the
    # ordered (bit-exact) tile kernel -- integer-valued sums exactly
    tables = synthetic.make_tables(12, n, m, strategy, leaves_per_tree=k, random_weights=(strategy == "branch"))
    dtab = dev.upload(tables)
    g = dtab.build()
    w = g.download()
    g.free()
    g = dtab.build(scatter=True)
    ws = g.download()
    g.free()
    dtab.free()
    if strategy in ("one", "depth"):
        assert np.array_equal(ws, w)  # integer addends: any order gives the same sum
        assert np.array_equal(ws, ws.T)
    else:
        scale = np.maximum(np.abs(w), 1e-300)
        assert float(np.max(np.abs(ws - w) / scale)) <= 1e-12
        assert np.array_equal(ws != 0, w != 0)
        # (the two triangles receive their adds in different orders: the atomic variant is not
        # even bitwise symmetric -- one more reason the product path is the ordered kernel)
        assert float(np.max(np.abs(ws - ws.T) / scale)) <= 1e-12
