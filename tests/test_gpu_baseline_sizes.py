"""Parity at BASELINE.json's full sizes (needs an MI355X): configs[2], [3] and [4].

What can be compared outright is: rows of W against the C oracle, bit for bit
(sampled -- the oracle is O(V M) per row); the embedding against scikit-learn's
own `spectral_embedding` on the device-built W (configs[2]: ~15 s of LAPACK on
the box; at 50 000 and 100 000 taxa the dense LU needs hours and >3 copies of a
20/80 GB matrix, so there the check is through size-independent properties:
symmetry, zero diagonal, the true residual of the returned vector recomputed on
the host from sampled rows of W, orthogonality to the trivial eigenvector).
Bars: W bit-exact; Fiedler entries within 1e-10 of scikit-learn on BOTH scales
(the embedding `maps` and the unit-norm eigenvector); labels identical.
"""

import os
import sys
import time
from pathlib import Path

import numpy as np
import pytest

from oracle import scs_oracle as so
from oracle import tables_oracle as to
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

pytestmark = pytest.mark.gpu

FIEDLER_TOL = 1e-10


@pytest.fixture(scope="module")
def dev():
    d = Device(0)
    yield d
    d.close()


def _sample_rows(n, count, seed):
    return np.unique(np.random.RandomState(seed).randint(0, n, size=count)).astype(np.int32)


def _check_rows_bit_exact(graph, tables, rows, row_begin=0):
    want = to.pcg_rows(tables, rows)
    bad = 0
    for i, r in enumerate(rows):
        got = graph.download_rows(int(r), 1)[0]
        bad += int(np.count_nonzero(got != want[i]))
    assert bad == 0, f"{bad} cells of {len(rows)} sampled rows differ from the oracle"
    return want


def _residual_from_rows(graph, rows, w_rows, deg, x, lam):
    """||(S x - lam x)[rows]|| / sqrt(fraction) : the true residual estimated from sampled rows
    of W (S = D^-1/2 W D^-1/2), x the unit-norm eigenvector."""
    dinv = 1.0 / np.sqrt(deg)
    sx = dinv[rows] * (w_rows @ (dinv * x))
    r = sx - lam * x[rows]
    return float(np.linalg.norm(r) * np.sqrt(len(x) / len(rows)))


def test_config2_full_parity(dev):
    """configs[2]: 10 000 taxa / 500 trees / branch -- everything the oracle can give."""
    from sklearn.cluster import k_means

    n, m = 10000, 500
    tables = synthetic.make_tables(0, n, m, "branch")
    dtab = dev.upload(tables)
    graph = dtab.build()
    try:
        rows = _sample_rows(n, 72, 11)
        assert len(rows) >= 64
        _check_rows_bit_exact(graph, tables, rows)
        w = graph.download()
        assert np.array_equal(w, w.T)
        assert not np.any(np.diag(w))
        rs = np.random.RandomState(0)
        v0 = rs.uniform(-1, 1, n)
        maps, stats = graph.fiedler(v0)
        _, labels, _ = k_means(maps, 2, random_state=rs, n_init=10, verbose=False)
    finally:
        graph.free()
        dtab.free()
    assert stats["converged"] == 1

    # scikit-learn on the same matrix, exactly as SpectralClustering.fit goes about it: the
    # embedding (draws the ARPACK start vector), then k_means on the same stream
    t0 = time.perf_counter()
    rs_ref = np.random.RandomState(0)
    ref = to.sign_flip_columns(so.spectral_maps(w, rs_ref))
    _, labels_ref, _ = k_means(ref, 2, random_state=rs_ref, n_init=10, verbose=False)
    t_ref = time.perf_counter() - t0

    _, dd = to.normalized_operator(w)
    err_maps = float(np.max(np.abs(maps[:, 1] - ref[:, 1])))
    err_unit = float(np.max(np.abs(maps[:, 1] * dd - ref[:, 1] * dd)))
    err_col0 = float(np.max(np.abs(maps[:, 0] - ref[:, 0])))
    mism = int(np.count_nonzero(labels != labels_ref))
    mism = min(mism, n - mism)  # label names are arbitrary
    gap = stats["lambda"][1] - stats["lambda_next"]
    print(f"CFG2 lambda2 {stats['lambda'][1]:.12f} lambda3 {stats['lambda_next']:.12f} gap {gap:.3e} "
          f"residual {stats['resid'][1]:.3e} iterations {stats['iterations']} err_maps {err_maps:.3e} "
          f"err_unit {err_unit:.3e} err_col0 {err_col0:.3e} labels_mismatched {mism} "
          f"sklearn {t_ref:.1f} s")
    assert err_maps <= FIEDLER_TOL
    assert err_unit <= FIEDLER_TOL
    assert err_col0 <= FIEDLER_TOL
    assert mism == 0


def test_config2_size_bootstrap_rows_bit_exact(dev):
    """configs[2]'s shape under the one weighting that is not monotone in the depth
    (`bootstrap`: the general tile kernel, scs_gen.h): sampled rows bit for bit, symmetry."""
    n, m = 10000, 500
    tables = synthetic.make_tables(0, n, m, "bootstrap")
    assert not tables.monotone
    dtab = dev.upload(tables)
    graph = dtab.build()
    try:
        assert graph.build_stats["n_batches"] > 1
        _check_rows_bit_exact(graph, tables, _sample_rows(n, 40, 3))
        w = graph.download()
        assert np.array_equal(w, w.T)
        assert not np.any(np.diag(w))
    finally:
        graph.free()
        dtab.free()


def _large_config_properties(dev, n, m, random_weights, n_rows, blocks, second_solve=False):
    tables = synthetic.make_tables(0, n, m, "branch", random_weights=random_weights)
    dtab = dev.upload(tables)
    graph = dtab.build()
    try:
        bstats = graph.build_stats
        rows = _sample_rows(n, n_rows, 5)
        w_rows = _check_rows_bit_exact(graph, tables, rows)
        # symmetry on sampled blocks: rows [a, a+k) against the transposed columns
        for a, b in blocks:
            k = 96
            ra = graph.download_rows(a, k)
            rb = graph.download_rows(b, k)
            assert np.array_equal(ra[:, b:b + k], rb[:, a:a + k].T)
            assert not np.any(np.diag(ra[:, a:a + k]))
        deg = graph.degrees()
        v0 = np.random.RandomState(0).uniform(-1, 1, n)
        maps, stats = graph.fiedler(v0)
        maps2 = stats2 = None
        if second_solve:
            # an independent solve: another start vector, the other block width -- another Krylov
            # sequence altogether, and (round 5, where the image is made: up to 16 GiB, configs[3]) the
            # other arithmetic: the default on one device is then width 4 with the loop's SYMM streaming
            # the single-precision image of W; width 8 streams W itself throughout
            maps2, stats2 = graph.fiedler(np.random.RandomState(1).uniform(-1, 1, n),
                                          block=8 if stats["block"] == 4 else 4)
    finally:
        graph.free()
        dtab.free()
    assert stats["converged"] == 1, stats
    assert np.all(deg > 0)
    # degrees of the sampled rows agree with the oracle rows' sums
    assert np.allclose(deg[rows], w_rows.sum(axis=1), rtol=1e-13, atol=0)
    dd = np.sqrt(deg)
    x = maps[:, 1] * dd
    nrm = float(np.linalg.norm(x))
    assert abs(nrm - 1.0) <= 1e-12  # maps = unit-norm eigenvector / sqrt(degree)
    x /= nrm
    u = dd / np.linalg.norm(dd)
    lam = stats["lambda"][1]
    res = _residual_from_rows(None, rows, w_rows, deg, x, lam)
    # column 0 is the trivial eigenvector: exactly constant (scikit-learn: to 1e-18)
    c0 = maps[:, 0]
    # sign rule: the entry of largest magnitude of each column is positive
    for c in range(2):
        assert maps[np.argmax(np.abs(maps[:, c])), c] > 0
    print(f"CFG n={n} m={m} tiles {bstats['n_tiles']} batches {bstats['n_batches']} "
          f"build {bstats['total_ms']:.1f} ms lambda2 {lam:.12f} lambda3 {stats['lambda_next']:.12f} "
          f"solver residual {stats['resid'][1]:.3e} host residual (sampled rows) {res:.3e} "
          f"iterations {stats['iterations']} <x,u> {float(x @ u):.2e}")
    assert res <= 1e-11
    assert abs(float(x @ u)) <= 1e-12
    assert float(np.max(c0) - np.min(c0)) <= 1e-15 * float(np.max(np.abs(c0)))
    if second_solve:
        # Without LAPACK at this size the sampled residual alone bounds an entry only through
        # the gap (1e-11 / 2.6e-5): two solves that share nothing but the matrix and agree to
        # 1e-11 on the unit-norm scale put the entries where the 1e-10 bar asks (the measured
        # distance to scikit-learn's own solve, 3.7e-12, is profiles/r03_config3_vs_sklearn_final.json)
        assert stats2["converged"] == 1 and {stats["block"], stats2["block"]} == {4, 8}, (stats, stats2)
        assert (stats["n_apply32"] > 0) == (n <= 60000) and stats2["n_apply32"] == 0, (stats, stats2)
        x2 = maps2[:, 1] * dd
        x2 /= float(np.linalg.norm(x2))
        agree_unit = float(np.max(np.abs(x - x2)))
        agree_maps = float(np.max(np.abs(maps[:, 1] - maps2[:, 1])))
        print(f"CFG n={n} two independent solves: b={stats['block']} {stats['iterations']} it, "
              f"b={stats2['block']} {stats2['iterations']} it, max |dx| unit-norm {agree_unit:.3e}, embedding {agree_maps:.3e}, "
              f"|dlambda2| {abs(stats['lambda'][1] - stats2['lambda'][1]):.2e}")
        assert agree_unit <= 1e-11
        assert agree_maps <= 1e-11
    return maps, stats


def test_config3_single_device_properties(dev):
    """configs[3] on ONE device: 50 000 taxa / 2 000 trees / branch (W = 20 GB)."""
    _large_config_properties(dev, 50000, 2000, False, 40, [(0, 49000), (12345, 30000)], second_solve=True)


def test_config3_two_ranks_match_single(dev):
    """configs[3] row-partitioned over two in-process ranks (the RCCL code path with the
    thread-barrier communicator) -- rows of W bit-exact on both ranks, same embedding on both."""
    from tests.test_gpu_parity import _run_local_group

    n, m = 50000, 2000
    tables = synthetic.make_tables(0, n, m, "branch")
    v0 = np.random.RandomState(0).uniform(-1, 1, n)
    splits = [0, 24960, n]

    import threading

    from spectralclustersupertree_amd import _native as nv

    lib = nv.load_library()
    group = nv.C.c_void_p()
    nv.check(lib.scs_local_group_create(2, nv.C.byref(group)))
    out, err = [None, None], [None, None]

    def worker(rank):
        try:
            d = Device(0, rank, 2, _local_group=group)
            dtab = d.upload(tables)
            g = dtab.build(splits[rank], splits[rank + 1], shared=True)
            rows = (splits[rank] + _sample_rows(splits[rank + 1] - splits[rank], 12, 3 + rank)).astype(np.int32)
            got = np.vstack([g.download_rows(int(r), 1) for r in rows])
            maps, stats = g.fiedler(v0)
            out[rank] = (rows, got, maps, stats, g.build_stats)
            g.free()
            dtab.free()
            d.close()
        except BaseException as e:  # noqa: BLE001
            err[rank] = e

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    lib.scs_local_group_destroy(group)
    assert err == [None, None], err
    for rank in range(2):
        rows, got, maps, stats, bstats = out[rank]
        assert bstats["symmetric"] == 2
        assert np.array_equal(got, to.pcg_rows(tables, rows)), rank
        assert stats["converged"] == 1
    assert np.array_equal(out[0][2], out[1][2])
    print("CFG3 2 ranks: lambda2", out[0][3]["lambda"][1], "iterations", out[0][3]["iterations"],
          "exchange ms", out[0][4]["exchange_ms"])
    _ = _run_local_group  # (shared helper kept importable)


def test_config3_two_ranks_upper_triangle_job(dev):
    """configs[3] as an SCS_BUILD_UPPER job of two in-process ranks: no tile exchange, each rank
    stores and streams its trapezoid of the upper triangle, the partial products are added in
    rank order -- rows bit-exact, the same embedding on both ranks, within 1e-10 of the
    single-device solve."""
    from spectralclustersupertree_amd.partition import LocalTeams, row_splits_upper

    n, m = 50000, 2000
    # (page-locked tables: each rank's upload returns after the first tree batch's worth of trees and
    # the rest arrives on that rank's copy stream while its build is under way -- include/scs_hip.h)
    tables = synthetic.make_tables(0, n, m, "branch", pinned=True)
    v0 = np.random.RandomState(0).uniform(-1, 1, n)
    dtab = dev.upload(tables)
    g = dtab.build()
    maps_single, stats_single = g.fiedler(v0)
    g.free()
    dtab.free()
    splits = row_splits_upper(n, 2)
    teams = LocalTeams(2)

    def rank_work(team):
        dtab = team.device.upload(tables)
        g = dtab.build(splits[team.rank], splits[team.rank + 1], upper=True)
        lo = splits[team.rank]
        rows = (lo + _sample_rows(splits[team.rank + 1] - lo, 10, 7 + team.rank)).astype(np.int32)
        got = np.vstack([g.download_rows(int(r), 1) for r in rows])
        maps, stats = g.fiedler(v0)
        bstats = g.build_stats
        g.free()
        dtab.free()
        return rows, got, maps, stats, bstats

    try:
        out = teams.run(rank_work)
    finally:
        teams.close()
    for rows, got, maps, stats, bstats in out:
        want = to.pcg_rows(tables, rows)
        for i, r in enumerate(rows):
            c0 = int(r) // 256 * 256
            assert np.array_equal(got[i, c0:], want[i, c0:])
        assert stats["converged"] == 1 and bstats["exchange_bytes"] == 0
        assert np.array_equal(maps, out[0][2])
        assert np.max(np.abs(maps - maps_single)) <= FIEDLER_TOL
    # half the streamed bytes of the row-partitioned solve: 4 V^2 over the job per application
    job_bytes = sum(o[3]["apply_bytes"] for o in out)
    assert job_bytes <= 1.1 * 4.0 * n * n + 1e9
    print("CFG3 upper job: build ms", [round(o[4]["total_ms"], 1) for o in out], "solve ms",
          [round(o[3]["solve_ms"], 1) for o in out], "iterations", out[0][3]["iterations"],
          "single-device solve ms", round(stats_single["solve_ms"], 1))


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("SCS_SLOW_TESTS"),
                    reason="about 6 minutes of LAPACK on the box's host cores: set SCS_SLOW_TESTS=1; "
                           "the committed run is profiles/r03_config3_vs_sklearn.json")
def test_config3_fiedler_against_the_reference_solve():
    """configs[3], measured rather than bounded: scikit-learn's own shift-invert ARPACK solve
    (the reference's call, scs.py:235-252) on the downloaded 20 GB matrix against
    ``scs_fiedler`` -- Fiedler column within 1e-10 on both scales, labels identical."""
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import reference_check_large

    res = reference_check_large.run(50000, 2000, False)
    print("CFG3 vs scikit-learn:", res)
    assert res["err_maps"] <= FIEDLER_TOL
    assert res["err_unit"] <= FIEDLER_TOL
    assert res["err_col0"] <= FIEDLER_TOL
    assert res["labels_mismatched"] == 0
    assert res["stream_position_equal"]


@pytest.mark.slow
def test_config4_single_device_properties(dev):
    """configs[4] on ONE device: 100 000 taxa / 5 000 weighted trees / branch (W = 80 GB).  As at
    configs[3] the 1e-10 bar rests on two solves that share nothing but the matrix (block widths
    8 and 4, different start vectors) agreeing within 1e-11 on the unit-norm scale -- the sampled
    residual alone bounds an entry only through the gap (1.1e-4 here)."""
    _large_config_properties(dev, 100000, 5000, True, 24, [(0, 99000)], second_solve=True)


@pytest.mark.slow
def test_config4_full_recursion_properties(dev):
    """configs[4]'s "full recursion" leg at its real shape -- 100 000 taxa / 5 000 weighted trees
    -- through ``construct_supertree``'s recursion (reference: scs.py:96-174): every taxon once,
    the root's subtrees = the top-level labels, every call partitions its taxa, and the same
    supertree and RandomState position as the committed run."""
    import json

    # (this module's own context still caches the 80 GB buffer of the test above: the recursion runs on the
    # process-wide context and needs the room -- level forests, look-ahead workers)
    dev.trim()

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import full_recursion_check

    res = full_recursion_check.run(100000, 5000, True)
    print("CFG4 full recursion:", res)
    assert res["every_taxon_exactly_once"]
    assert res["top_level_parts_equal_top_level_labels"]
    assert res["every_call_partitions_its_taxa"]
    committed = Path(__file__).resolve().parents[1] / "profiles" / "r03_config4_full_recursion_run7_final.json"
    if committed.exists():
        want = json.loads(committed.read_text())
        assert res["newick_sha256"] == want["newick_sha256"]
        assert res["random_state_next_draw"] == want["random_state_next_draw"]
        assert res["spectral_calls"] == want["spectral_calls"]
    # (round 6) the walk below 2 048 taxa is level-synchronous with provisional labels: every one of them was
    # confirmed or repaired against the true draws (the digest above), and no solve took the warn-and-accept
    # branch of scs._fiedler_checked (ACCEPT_RESIDUAL)
    assert res["level_engine"]["roots"] >= 1
    assert res["accepted_residual_warnings"] == 0


def test_page_locked_tables_arrive_behind_the_first_tree_batch(dev):
    """scs_tables_upload with page-locked arrays returns after the first tree batch's worth of
    trees; the rest travels on the copy stream while scs_pcg_build works on the first batches
    (include/scs_hip.h).  Same W, bit for bit, as from pageable arrays (everything copied
    first) -- both tile kernels, and a second build from the same tables."""
    from spectralclustersupertree_amd import _native as nv

    n, m = 10000, 500
    for strategy in ("branch", "bootstrap"):
        pinned = synthetic.make_tables(0, n, m, strategy, pinned=True)
        plain = synthetic.make_tables(0, n, m, strategy)
        assert np.array_equal(pinned.leaf_taxon, plain.leaf_taxon) and np.array_equal(pinned.adj_val, plain.adj_val)
        rows = _sample_rows(n, 24, 7)
        got = []
        for tables in (pinned, plain):
            dtab = dev.upload(tables)
            graph = dtab.build()
            try:
                # (pageable monotone tables: ONE batch of 500 trees for the producer / consumer kernel;
                # page-locked ones: the 64 trees that have arrived, then the rest)
                if tables is pinned or strategy == "bootstrap":
                    assert graph.build_stats["n_batches"] > 1
                got.append(_check_rows_bit_exact(graph, tables, rows))
                if tables is pinned:  # the tables are complete now: a second build reads them as they are
                    again = dtab.build()
                    try:
                        assert np.array_equal(again.download_rows(4000, 64), graph.download_rows(4000, 64))
                    finally:
                        again.free()
            finally:
                graph.free()
                dtab.free()
        assert np.array_equal(got[0], got[1])
    # a taxon id out of range in the LAST tree: the upload cannot see it any more, the build reports it
    bad = synthetic.make_tables(0, n, m, "branch", pinned=True)
    bad.leaf_taxon[-5] = n + 3
    dtab = dev.upload(bad)
    try:
        with pytest.raises(nv.ScsError, match="out of range") as err:
            dtab.build()
        assert err.value.code == nv.EINVAL
    finally:
        dtab.free()
    # freed without ever being built: the pending copies are drained first
    dev.upload(synthetic.make_tables(1, n, m, "branch", pinned=True)).free()
    dev.synchronize()
