"""The multi-rank path behind the public API (needs an MI355X): several ranks walking one
recursion, nodes above the threshold solved collectively on group-aligned row splits
(SURVEY.md 8e, 8f-1), children below it on single devices.  The ranks here are threads of
one process on one GPU (libscs_hip's in-process communicator stands in for RCCL; the
collective code path above it is the same) -- a real job has one process per GPU."""

import numpy as np
import pytest
from reference_cases import DATA_DIR, FILE_CASES

from oracle import scs_oracle as so
from oracle import tables_oracle as to
from spectralclustersupertree_amd import construct_supertree, synthetic
from spectralclustersupertree_amd import flatten as fl
from spectralclustersupertree_amd.backend import Device
from spectralclustersupertree_amd.load import load_tree_arrays
from spectralclustersupertree_amd.partition import LocalTeams, group_splits, row_splits
from spectralclustersupertree_amd.scs import relabel_for_contraction, spectral_bipartition_device
from spectralclustersupertree_amd.tree import TreeNode, load_tree

pytestmark = pytest.mark.gpu


def _with_twins(tree: TreeNode, twinned: set[str]) -> TreeNode:
    """A copy of the tree in which every tip named in `twinned` is a cherry (x, x_twin): the
    two always travel together, so contraction merges them."""
    if tree.is_tip():
        if tree.name in twinned:
            return TreeNode(None, [TreeNode(tree.name, None, 0.01), TreeNode(tree.name + "_twin", None, 0.02)],
                            tree.length, 90.0)
        return TreeNode(tree.name, None, tree.length, tree.support)
    return TreeNode(tree.name, [_with_twins(c, twinned) for c in tree.children], tree.length, tree.support)


def _twin_tables(n, m, k, n_twins, strategy="branch"):
    trees = synthetic.tree_objects(21, n, m, leaves_per_tree=k)
    rs = np.random.RandomState(2)
    twinned = {synthetic.taxon_name(int(i)) for i in rs.choice(n, size=n_twins, replace=False)}
    trees = [_with_twins(t, twinned) for t in trees]
    names = sorted(so._all_tips(trees))
    weights = [1.0 + 0.25 * (i % 3) for i in range(m)]
    return fl.flatten_trees(trees, weights, strategy, names), trees, weights


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_build_contract_fiedler_matches_single_rank(world):
    # build -> contract -> fiedler on group-aligned row splits vs the single-rank result
    # (reference: scs.py:261-387 contraction, then :210-258)
    tables, _, _ = _twin_tables(640, 14, 600, 150)
    assert tables.monotone
    groups = fl.contraction_groups(tables)
    n_groups = int(groups.max()) + 1
    assert n_groups < tables.n_taxa - 100  # the twins (present in every tree they share) merge
    work, perm, group_start = relabel_for_contraction(tables, groups)
    splits = row_splits(tables.n_taxa, world, group_start)
    gsp = group_splits(splits, group_start)

    with Device(0) as dev:
        dtab = dev.upload(work)
        g = dtab.build().contract(group_start)
        w_single = g.download()
        v0 = np.random.RandomState(0).uniform(-1, 1, n_groups)
        maps_single, _ = g.fiedler(v0)
        g.free()
        dtab.free()
    w_ref, _ = to.pcg_dense(work)
    assert np.array_equal(w_single, to.contract_dense(w_ref, group_start))

    teams = LocalTeams(world)

    def rank_work(team):
        dtab = team.device.upload(work)
        g = dtab.build(splits[team.rank], splits[team.rank + 1], shared=True).contract(group_start)
        assert g.shape == (n_groups, gsp[team.rank], gsp[team.rank + 1])
        w = g.download()
        maps, stats = g.fiedler(v0)
        g.free()
        dtab.free()
        return w, maps, stats

    try:
        out = teams.run(rank_work)
    finally:
        teams.close()
    assert np.array_equal(np.vstack([o[0] for o in out]), w_single)
    for w, maps, stats in out:
        assert stats["converged"] == 1
        assert np.array_equal(maps, out[0][1])
        assert np.max(np.abs(maps - maps_single)) <= 1e-10


@pytest.mark.parametrize(("world", "n", "kind"), [(2, 4608, "branch"), (4, 6000, "depth"), (3, 5000, "planted")])
def test_row_partitioned_job_streams_the_image_of_its_rows(world, n, kind, monkeypatch):
    """Round 6: the mixed-precision loop (single-precision image of W for the search directions, S X / S P
    renewed and the result confirmed through W) in the row-partitioned layout north_star names -- every rank
    keeps the image of ITS rows, k_symm streams it.  The embedding of 2 / 3 / 4 in-process ranks stays within
    1e-10 of one device AND of scikit-learn (the reference's own call, scs.py:235-252), every rank ran on the
    image, all ranks hold the same bits."""
    from oracle import scs_oracle as so

    monkeypatch.delenv("SCS_LOWP", raising=False)
    if kind == "planted":
        tables = synthetic.make_tables(n, n, 12, "branch", random_weights=True, planted_spr=int(np.ceil(0.02 * n)))
    else:
        tables = synthetic.make_tables(n, n, 20, kind, random_weights=(kind == "branch"))
    v0 = np.random.RandomState(1).uniform(-1, 1, n)
    with Device(0) as dev:
        dtab = dev.upload(tables)
        g = dtab.build()
        w = g.download()
        maps_single, stats_single = g.fiedler(v0)
        g.free()
        dtab.free()
    assert stats_single["n_apply32"] > 0
    splits = row_splits(n, world, None)
    teams = LocalTeams(world)

    def rank_work(team):
        dtab = team.device.upload(tables)
        g = dtab.build(splits[team.rank], splits[team.rank + 1], shared=True)
        maps, stats = g.fiedler(v0)
        again, stats2 = g.fiedler(v0)  # (the image is there already; the degrees are not gathered again)
        g.free()
        dtab.free()
        return maps, stats, again, stats2

    try:
        out = teams.run(rank_work)
    finally:
        teams.close()
    ref = to.sign_flip_columns(so.spectral_maps(w, np.random.RandomState(1)))
    _, dd = to.normalized_operator(w)
    rows = [splits[r + 1] - splits[r] for r in range(world)]
    for r, (maps, stats, again, stats2) in enumerate(out):
        assert stats["converged"] == 1 and stats["block"] == 4
        assert stats["n_apply32"] > 0 and stats2["n_apply32"] > 0, stats
        # per rank and launch: 4 bytes a cell of its rows, half of what the double-precision launch streams
        assert stats["apply32_bytes"] < 0.55 * stats["apply_bytes"]
        assert abs(stats["apply32_bytes"] - (4.0 * rows[r] * n + 8.0 * n * 4 + 8.0 * rows[r] * 4)) <= 1.0
        assert np.array_equal(maps, out[0][0]) and np.array_equal(again, maps)
        assert float(np.max(np.abs((maps[:, 1] - maps_single[:, 1]) * dd))) <= 1e-10
        assert float(np.max(np.abs(maps[:, 1] - ref[:, 1]))) <= 1e-10
        assert float(np.max(np.abs((maps[:, 1] - ref[:, 1]) * dd))) <= 1e-10


@pytest.mark.parametrize("world", [2, 3])
def test_bipartition_through_a_team_equals_single_device(world):
    tables, _, _ = _twin_tables(500, 12, 470, 90, strategy="depth")
    with Device(0) as dev:
        want_members, want_labels = spectral_bipartition_device(
            tables, np.random.RandomState(5), contract_edges=True, device=dev)
    teams = LocalTeams(world, shard_min=100)

    def rank_work(team):
        report = {}
        rs = np.random.RandomState(5)
        members, labels = spectral_bipartition_device(tables, rs, contract_edges=True, team=team,
                                                      report=report)
        return members, labels, report, rs.randint(1 << 30)

    try:
        out = teams.run(rank_work)
    finally:
        teams.close()
    for members, labels, report, draw in out:
        assert report["sharded"] and len(report["splits"]) == world + 1
        assert [m.tolist() for m in members] == [m.tolist() for m in want_members]
        assert np.array_equal(labels, want_labels)
        assert draw == out[0][3]  # every rank left the stream in the same state


def test_construct_supertree_with_a_team_shared_stream():
    # the whole recursion walked by two ranks: big nodes collectively, small ones on each
    # rank's own device with the shared stream -> the single-device result, exactly
    tables, trees, weights = _twin_tables(420, 10, 400, 60)
    want = construct_supertree(trees, weights, "branch", random_state=np.random.RandomState(1))
    teams = LocalTeams(2, shard_min=120)
    try:
        out = teams.run(lambda team: construct_supertree(trees, weights, "branch",
                                                         random_state=np.random.RandomState(1), team=team))
    finally:
        teams.close()
    for got in out:
        assert got.sorted().get_newick() == want.sorted().get_newick()


@pytest.mark.parametrize("world, n, m, leaves, strategy, contract, shard_min", [
    (2, 1500, 40, None, "branch", True, 300),  # (nothing contracts at the root: children of 672 / 828 vertices)
    (3, 2000, 10, 1500, "bootstrap", True, 450),  # (1 662 groups of 2 000 taxa: group-aligned splits, 696 / 966)
    (4, 1100, 16, None, "depth", False, 300),
])
def test_level_engine_under_a_team_deals_the_larger_nodes(world, n, m, leaves, strategy, contract, shard_min):
    """Several ranks, ONE stream, the level engine on every rank: the larger nodes of every level are dealt over the
    ranks (those of shard_min vertices and more solved by all of them together), the embeddings exchanged -- and the
    tree, and the position the stream is left at, are the single device's (reference: scs.py:158-166, one RandomState
    through every node).  Held against the same team walking every node by itself (rounds 2-5) as well."""
    from spectralclustersupertree_amd import levels, scs

    kw = {} if leaves is None else {"leaves_per_tree": leaves}
    # (one copy of the input per rank: a rank keeps its own resident forest on the arrays it walks)
    copies = [synthetic.tree_arrays(n + m, n, m, random_weights=True, **kw) for _ in range(world + 1)]

    def run(team):
        rs = np.random.RandomState(7)
        arrays = copies[world if team is None else team.rank]
        tree = scs._construct(arrays, strategy, contract, rs, team=team)
        return tree.get_newick(), int(rs.randint(1 << 30))

    want = run(None)
    single = dict(levels.stats)
    assert single["roots"] >= 1 and single["n_large"] >= 3  # the engine ran, with nodes to deal
    assert single["team_dealt"] == 0 and single["team_collective"] == 0
    teams = LocalTeams(world, shard_min=shard_min)
    try:
        out = teams.run(run)
    finally:
        teams.close()
    st = dict(levels.stats)
    for got in out:
        assert got == want
    assert st["roots"] >= 1 and st["team_dealt"] > 0 and st["team_received"] > 0 and st["team_collective"] > 0
    copies = [synthetic.tree_arrays(n + m, n, m, random_weights=True, **kw) for _ in range(world + 1)]
    teams = LocalTeams(world, shard_min=shard_min)
    for t in teams.teams:
        t.level_engine = False
    try:
        out = teams.run(run)
    finally:
        teams.close()
    assert levels.stats["roots"] == 0  # (node by node on every rank)
    for got in out:
        assert got == want


def test_construct_supertree_with_a_team_children_one_per_device():
    # "forked" streams: sibling sub-problems below the threshold are dealt to the ranks and the
    # subtrees exchanged; the reference's fixture has a seed-independent answer
    name, src, exp, weighting = next(c for c in FILE_CASES if "supertriplets" in c[0])
    arrays = load_tree_arrays(DATA_DIR / src)
    expected = load_tree(DATA_DIR / exp)
    teams = LocalTeams(3, shard_min=60, child_rng="forked")
    try:
        out = teams.run(lambda team: construct_supertree(arrays, pcg_weighting=weighting,
                                                         random_state=np.random.RandomState(0), team=team))
    finally:
        teams.close()
    for got in out:
        assert got.sorted().same_shape(expected.sorted())
    assert len({g.sorted().get_newick() for g in out}) == 1


# ---------------------------------------------------------------------------
# SCS_BUILD_UPPER: the job keeps only the upper triangle -- no exchange, half the bytes
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("world,strategy,n", [(2, "branch", 1300), (3, "bootstrap", 1700), (4, "branch", 2100)])
def test_upper_triangle_job_build_apply_fiedler(world, strategy, n):
    from spectralclustersupertree_amd.partition import row_splits_upper

    tables = synthetic.make_tables(31 + world, n, 18, strategy, leaves_per_tree=n - 57, random_weights=True)
    w_ref, _ = to.pcg_dense(tables)
    s_ref, _ = to.normalized_operator(w_ref)
    splits = row_splits_upper(n, world)
    assert all(s % 256 == 0 for s in splits[:-1]) and splits[-1] == n
    x = np.random.RandomState(1).standard_normal((n, 4))
    v0 = np.random.RandomState(0).uniform(-1, 1, n)
    with Device(0) as dev:
        dtab = dev.upload(tables)
        g = dtab.build()
        maps_single, _ = g.fiedler(v0)
        g.free()
        dtab.free()

    teams = LocalTeams(world)

    def rank_work(team):
        dtab = team.device.upload(tables)
        g = dtab.build(splits[team.rank], splits[team.rank + 1], upper=True)
        w = g.download()
        deg = g.degrees()
        y = g.apply(x)
        maps, stats = g.fiedler(v0)
        bstats = g.build_stats
        with pytest.raises(Exception, match="UPPER"):
            g.contract(np.arange(n + 1, dtype=np.int32))
        g.free()
        dtab.free()
        return w, deg, y, maps, stats, bstats

    try:
        out = teams.run(rank_work)
    finally:
        teams.close()
    cells = 0
    for r, (w, deg, y, maps, stats, bstats) in enumerate(out):
        lo, hi = splits[r], splits[r + 1]
        for i in range(lo, hi):
            c0 = i // 256 * 256  # a row's stored cells start with its 256-column diagonal tile
            assert np.array_equal(w[i - lo, c0:], w_ref[i, c0:]), (r, i)
            assert not np.any(w[i - lo, :c0])
        cells += bstats["cell_trees"]
        assert np.allclose(deg, w_ref.sum(axis=1)[lo:hi], rtol=1e-13, atol=0)
        assert np.max(np.abs(y - (s_ref @ x)[lo:hi])) <= 1e-13
        assert stats["converged"] == 1
        assert np.array_equal(maps, out[0][3])  # the same bits on every rank
        assert np.max(np.abs(maps - maps_single)) <= 1e-10
    # every cell of the upper triangle evaluated about once across the job (whole tiles)
    assert cells <= 0.5 * n * n * tables.n_trees * 1.6


def test_bipartition_through_a_team_takes_the_upper_triangle_when_nothing_contracts(monkeypatch):
    # (60 full trees: no two taxa share a root side in every one of them, nothing contracts)
    monkeypatch.setenv("SCS_MULTI_MODE", "upper")  # (round 5: the default is the row-partitioned layout)
    tables = synthetic.make_tables(5, 1100, 60, "depth")
    assert int(fl.contraction_groups(tables).max()) + 1 == tables.n_taxa
    with Device(0) as dev:
        want_members, want_labels = spectral_bipartition_device(
            tables, np.random.RandomState(5), contract_edges=True, device=dev)
    teams = LocalTeams(2, shard_min=100)

    def rank_work(team):
        report = {}
        rs = np.random.RandomState(5)
        members, labels = spectral_bipartition_device(tables, rs, contract_edges=True, team=team, report=report)
        return members, labels, report, rs.randint(1 << 30)

    try:
        out = teams.run(rank_work)
    finally:
        teams.close()
    for members, labels, report, draw in out:
        assert report["sharded"] and report["upper"] and report["build"]["exchange_bytes"] == 0
        assert [m.tolist() for m in members] == [m.tolist() for m in want_members]
        assert np.array_equal(labels, want_labels)
        assert draw == out[0][3]


def test_upper_triangle_job_equals_the_row_partitioned_job(monkeypatch):
    """The two multi-rank layouts against each other (two in-process ranks, nothing contracts):
    the stored cells of the upper-triangle job are the row-partitioned job's, bit for bit, the
    embeddings agree within the Fiedler tolerance and both give the single-device labels;
    SCS_MULTI_MODE=upper / shared select the layout (default since round 5: shared)."""
    from spectralclustersupertree_amd.partition import row_splits_upper

    n = 1500
    tables = synthetic.make_tables(77, n, 24, "branch", random_weights=True)
    v0 = np.random.RandomState(0).uniform(-1, 1, n)
    sp_rows, sp_upper = row_splits(n, 2), row_splits_upper(n, 2)
    teams = LocalTeams(2, shard_min=100)

    def rank_work(team):
        dtab = team.device.upload(tables)
        got = {}
        for mode, splits in (("shared", sp_rows), ("upper", sp_upper)):
            g = dtab.build(splits[team.rank], splits[team.rank + 1], shared=(mode == "shared"),
                           upper=(mode == "upper"))
            maps, stats = g.fiedler(v0)
            got[mode] = (g.download(), maps, stats, splits[team.rank])
            g.free()
        dtab.free()
        return got

    def rank_bipartition(team):
        rep = {}
        _, labels = spectral_bipartition_device(tables, np.random.RandomState(5), contract_edges=False,
                                                team=team, report=rep)
        return labels, rep

    try:
        out_got = teams.run(rank_work)
        by_mode = {}
        for mode in ("upper", "shared"):
            monkeypatch.setenv("SCS_MULTI_MODE", mode)
            by_mode[mode] = teams.run(rank_bipartition)
        monkeypatch.delenv("SCS_MULTI_MODE")
        by_mode["default"] = teams.run(rank_bipartition)
    finally:
        teams.close()
    out = [(out_got[r], {mode: by_mode[mode][r] for mode in by_mode}) for r in range(2)]
    with Device(0) as dev:
        _, want_labels = spectral_bipartition_device(tables, np.random.RandomState(5), contract_edges=False,
                                                     device=dev)
    for got, reports in out:
        w_s, maps_s, stats_s, lo_s = got["shared"]
        w_u, maps_u, stats_u, lo_u = got["upper"]
        assert stats_s["converged"] == 1 and stats_u["converged"] == 1
        # rows both layouts store on this rank
        for i in range(max(lo_s, lo_u), min(lo_s + w_s.shape[0], lo_u + w_u.shape[0])):
            c0 = i // 256 * 256
            assert np.array_equal(w_u[i - lo_u, c0:], w_s[i - lo_s, c0:])
        assert np.max(np.abs(maps_u - maps_s)) <= 1e-10
        assert reports["upper"][1]["upper"] and not reports["shared"][1]["upper"]
        assert not reports["default"][1]["upper"] and reports["default"][1]["sharded"]  # the default layout
        assert np.array_equal(reports["default"][0], want_labels)
        for mode in ("upper", "shared"):
            assert reports[mode][1]["sharded"]
            assert np.array_equal(reports[mode][0], want_labels)
