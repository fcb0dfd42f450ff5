"""``construct_supertree``: the reference's public entry point over the HIP core.

Same signature, defaults, error messages and recursion as the reference
(reference: src/sc_supertree/scs.py:18-174).  What changed is where the work of
one recursion node happens:

* proper-cluster-graph weights  -> ``scs_pcg_build`` (HIP, dense fp64 W in HBM),
* contraction                   -> host finds the groups, ``scs_graph_contract``
                                   max-reduces W on the device,
* spectral embedding            -> ``scs_fiedler`` (hand-written LOBPCG),
* label assignment              -> scikit-learn's ``k_means`` on the V x 2
  embedding, called on the host exactly as ``SpectralClustering.fit`` calls it
  (sklearn/cluster/_spectral.py:759-766) with the same ``RandomState`` stream
  position (the ARPACK start vector is still drawn first,
  sklearn/utils/_arpack.py:31-33), so labels match the reference's.

Tree restriction, tie-breaking and assembly stay on the host (north_star) -- but
not on Python tree objects: the source trees are converted once into flat node
arrays (``treearrays.TreeArrays``) and every level of the recursion restricts and
flattens them in ``libscs_host.so``.  The
object-walking recursion of the reference is kept as ``_construct_objects`` for
the equivalence tests.
There is no CPU fallback for the device steps: without libscs_hip.so and a
HIP device this function raises.
"""

from __future__ import annotations

import os
import threading
from collections.abc import Sequence

import numpy as np

from spectralclustersupertree_amd import _env
from spectralclustersupertree_amd import flatten as fl
from spectralclustersupertree_amd.backend import DEFAULT_MAX_ITER, DEFAULT_TOL, Device
from spectralclustersupertree_amd.treearrays import TreeArrays
from spectralclustersupertree_amd.tree import (
    TreeNode,
    connect_trees,
    is_not_completed,
    tip_names_to_tree,
)

_default_device: Device | None = None
_default_team = None
_default_team_checked = False

# Diagnostic hook: a list here receives one ``(member taxon names per vertex, labels)`` entry
# per spectral call of the recursion, in call order (``trace_nodes`` sets and clears it); the
# whole-recursion parity tests compare it with the oracle's trace node by node.
_node_trace: list | None = None
_last_ahead_stats: dict | None = None  # jobs of the latest recursion's ahead.Ahead (diagnostics, tests)
_traced_maps = None  # the embedding of the spectral call being traced (set by _labels_and_members)


class trace_nodes:
    """``with trace_nodes() as trace: construct_supertree(...)`` -- every spectral call of the
    recursion appends ``{"vertices": [tuple of taxon names, ...], "labels": int array, "maps": the
    V x 2 embedding the labels were assigned on}``."""

    def __enter__(self):
        global _node_trace
        _node_trace = []
        return _node_trace

    def __exit__(self, *exc):
        global _node_trace
        _node_trace = None
        return False


def default_device() -> Device:
    """Process-wide single-rank context, created on first use: GPU ``SCS_DEVICE`` (default 0),
    or this rank's single-device context when the process belongs to a launched job."""
    global _default_device
    team = default_team()
    if team is not None:
        return team.solo
    if _default_device is None:
        import os

        _default_device = Device(int(os.environ.get("SCS_DEVICE", "0")))
    return _default_device


def default_team():
    """The team of a job launched with one process per GPU (``torch.distributed.run`` sets
    RANK / WORLD_SIZE / LOCAL_RANK): created on first use, None in a plain run."""
    global _default_team, _default_team_checked
    if not _default_team_checked:
        import os

        # SCS_TEAM=0 (or off / none): never join the launcher's job -- for a process that
        # merely lives inside someone else's torch.distributed.run and wants one GPU to itself
        if os.environ.get("SCS_TEAM", "env").lower() in ("0", "off", "none", "no"):
            _default_team = None
        else:
            from spectralclustersupertree_amd.partition import team_from_env

            _default_team = team_from_env()
        _default_team_checked = True
    return _default_team


def _make_result_tree(newick: str):
    try:
        from cogent3 import make_tree  # type: ignore[import-not-found]
    except ImportError:
        from spectralclustersupertree_amd.tree import make_tree
    return make_tree(newick)


def relabel_for_contraction(tables: fl.TreeTables, groups: np.ndarray):
    """Renumber taxa so that every contraction group is a consecutive id range.

    Returns (tables in the new numbering, perm with perm[new] = old,
    group_start).  Groups are ordered by their smallest old id, members by old
    id, which is also the order of the reference's sorted merged-vertex tuples.
    """
    n = tables.n_taxa
    order = np.lexsort((np.arange(n), groups))  # by group, then old id
    new_of_old = np.empty(n, dtype=np.int32)
    new_of_old[order] = np.arange(n, dtype=np.int32)
    counts = np.bincount(groups, minlength=int(groups.max()) + 1)
    group_start = np.zeros(len(counts) + 1, dtype=np.int32)
    np.cumsum(counts, out=group_start[1:])
    relabelled = fl.TreeTables(
        n_taxa=n,
        tree_off=tables.tree_off,
        leaf_taxon=new_of_old[tables.leaf_taxon].astype(np.int32),
        adj_depth=tables.adj_depth,
        adj_val=tables.adj_val,
        tree_w=tables.tree_w,
        taxa=None,
        monotone=tables.monotone,
    )
    res = getattr(tables, "resident", None)
    if res is not None:  # the device copy follows: forest id -> node id -> contracted numbering
        forest, lut = res
        relabelled.resident = (forest, new_of_old if lut is None else new_of_old[lut].astype(np.int32))
    return relabelled, order.astype(np.int32), group_start


# A Ritz vector this far from an eigenvector is still good enough to cluster (the k-means
# split moves only when entries move by far more); above it the node is refused, as the
# reference's ARPACK call would raise ArpackNoConvergence rather than return a guess.
ACCEPT_RESIDUAL = 1e-8
ACCEPT_RESIDUAL_OVER_GAP = 1e-4  # and residual / (lambda2 - lambda3) at most this


def _fiedler_checked(graph, v0, tol, max_iter, block):
    """``graph.fiedler`` with the convergence contract of the recursion: an unconverged block
    is retried once with the widest block and four times the iterations; a residual that is
    then still above ``tol`` but below ACCEPT_RESIDUAL is used with a warning, anything
    worse raises (reference behaviour: scipy's eigsh raises, scs.py:252 propagates)."""
    import warnings

    from spectralclustersupertree_amd._native import ConvergenceError

    try:
        return graph.fiedler(v0, tol=tol, max_iter=max_iter, block=block)
    except ConvergenceError as first:
        try:
            # (the library clamps the width to what V supports: 3 b + 1 <= V; an upper-triangle
            # job's symmetric SYMM comes in widths 4 and 8 -- ask for what the job has)
            widest = 8 if getattr(graph, "upper", False) else 16
            return graph.fiedler(v0, tol=tol, max_iter=4 * max_iter, block=widest)
        except ConvergenceError as second:
            best = min((first, second), key=lambda e: max(e.stats["resid"]))
            resid = max(best.stats["resid"])
            # eigenvector error ~ residual / gap: the residual is accepted against the gap to
            # the next eigenvalue, not in absolute terms (lambda2 ~ lambda3 can turn a 1e-8
            # residual into a mix of the two eigenvectors)
            gap = abs(best.stats["lambda"][1] - best.stats["lambda_next"])
            if not (resid <= ACCEPT_RESIDUAL and resid <= ACCEPT_RESIDUAL_OVER_GAP * gap):
                msg = (f"Fiedler solve did not converge: residual {resid:.3e} "
                       f"(V = {best.stats['n_vertices']}, lambda2 {best.stats['lambda'][1]:.12g}, "
                       f"next {best.stats['lambda_next']:.12g})")
                raise RuntimeError(msg) from second
            warnings.warn(
                f"Fiedler solve stopped at residual {resid:.3e} (target {tol:.1e}, gap {gap:.3e}); "
                "clustering the block it reached", RuntimeWarning, stacklevel=3)
            best.stats["accepted_residual"] = resid
            return best.maps, best.stats


def spectral_bipartition_device(
    tables: fl.TreeTables,
    random_state: np.random.RandomState,
    *,
    contract_edges: bool,
    device: Device | None = None,
    team=None,
    tol: float = DEFAULT_TOL,
    max_iter: int = DEFAULT_MAX_ITER,
    block: int = 0,
    report: dict | None = None,
    presolved=None,
):
    """One recursion node's device work: build, contract, Fiedler, labels.

    Returns ``(groups, labels)``: the member taxa (ids of ``tables``) of every
    vertex in canonical order, and the 0/1 label of every vertex
    (reference: scs.py:125-134, 210-258).

    With a ``team`` of several ranks (one process per GPU, every rank making this call with
    the same arguments) a node of at least ``team.shard_min`` vertices is solved collectively:
    W row-partitioned on group-aligned splits, the upper-triangle tiles shared, one RCCL
    all-gather of the Krylov block per iteration; every rank receives the whole embedding and
    draws the same labels.  Smaller nodes run on the rank's own single-device context.

    A node of at most 128 taxa goes through ``scs_small_solve`` (three launches, batched);
    ``presolved`` carries the embedding of a node whose device work was already done in a
    batch with its siblings, or queued ahead of the walk (``_construct``, ``ahead.Ahead``).

    The embedding depends on the node's forest only: the reference's ARPACK start vector is
    drawn from ``random_state`` (the stream stays where the reference has it) and not used.
    """
    n = tables.n_taxa
    if presolved is not None:
        work, perm, group_start, n_groups, maps = presolved
        # the reference's ARPACK start vector is still the first draw from the stream
        random_state.uniform(-1, 1, n_groups)
        stats = None
        if isinstance(maps, (_Pending, _Begun)):
            maps, stats = maps.fetch(device or (team.solo if team is not None else default_device()))
        if report is not None:
            report.update(stats or {"n_vertices": n_groups, "block": 0})
            report["presolved"] = True
        return _labels_and_members(maps, random_state, n, perm, group_start, n_groups)

    work, perm, group_start, n_groups = prepare_node(tables, contract_edges)

    sharded = team is not None and team.world > 1 and n_groups >= team.shard_min
    splits = None
    upper = False
    if sharded:
        from spectralclustersupertree_amd.partition import row_splits, row_splits_upper

        try:
            if group_start is None and os.environ.get("SCS_MULTI_MODE", "shared") == "upper":
                # SCS_MULTI_MODE=upper, nothing contracts: the job keeps only the upper triangle of the
                # symmetric matrix -- no tile exchange, half the bytes per operator application.
                # The DEFAULT since round 5 is the row-partitioned layout BASELINE.json's north_star
                # names (whole rows on every rank, the Krylov block's slices all-gathered: the smaller
                # collective) until a run on N > 1 real GPUs has shown the upper-triangle job's rows
                # bit-equal there (bench.py --gpus N times both and prints both parities)
                splits = row_splits_upper(n, team.world)
                upper = True
            else:
                splits = row_splits(n, team.world, group_start)
        except ValueError:
            sharded = False  # fewer groups (or 256-row blocks) than ranks: every rank solves it alone
    if sharded:
        dev = team.device
    elif team is not None:
        dev = team.solo
    else:
        dev = device or default_device()

    if not sharded and n <= dev.SMALL_MAX_TAXA and tol == DEFAULT_TOL and block == 0 and _small_path():
        # a small node: tables -> W -> contraction -> Jacobi -> embedding in ONE launch
        # (scs_small_solve)
        random_state.uniform(-1, 1, n_groups)  # the stream position of the reference
        maps, lam = dev.small_solve([(work, group_start)])[0]
        if report is not None:
            report.update({"n_vertices": n_groups, "block": 0, "iterations": 0, "converged": 1,
                           "lambda": [float(lam[0]), float(lam[1])], "lambda_next": float(lam[2]),
                           "sharded": False, "splits": None, "small_path": True})
        return _labels_and_members(maps, random_state, n, perm, group_start, n_groups)

    # the reference's ARPACK start vector is the first draw from the stream
    random_state.uniform(-1, 1, n_groups)
    span = (splits[team.rank], splits[team.rank + 1]) if sharded else None
    maps, stats = _solve_node(dev, work, group_start, tol, max_iter, block, span, upper)
    if report is not None:
        report.update(stats)
        report["sharded"] = bool(sharded)
        report["upper"] = bool(sharded and upper)
        report["splits"] = splits
    return _labels_and_members(maps, random_state, n, perm, group_start, n_groups)


def _solve_node(dev, work, group_start, tol=DEFAULT_TOL, max_iter=DEFAULT_MAX_ITER, block=0, span=None,
                upper=False):
    """Tables -> W -> contraction -> embedding of one node on ``dev``: ``(maps, stats)``.
    ``span``: this rank's rows of a sharded job (``upper``: the upper-triangle job)."""
    dtab = dev.upload(work)
    try:
        if span is not None and upper:
            graph = dtab.build(span[0], span[1], upper=True)
        elif span is not None:
            graph = dtab.build(span[0], span[1], shared=True)
        else:
            graph = dtab.build()
    finally:
        dtab.free()
    try:
        if group_start is not None:
            graph = graph.contract(group_start)
        maps, stats = _fiedler_checked(graph, None, tol, max_iter, block)
        stats = dict(stats)
        stats["build"] = graph.build_stats
    finally:
        # (W, its image and the build's scratch go back to the device's arena here: the level forests and the
        # other contexts on this GPU carve their blocks out of the same memory -- csrc/scs_arena.h.  Nothing is
        # handed back to the driver during a recursion; Device.trim() does that)
        graph.free()
    return maps, stats



class _Begun:
    """The embedding of a small node whose ``scs_small_solve_begin`` is under way."""

    __slots__ = ("ticket",)

    def __init__(self, ticket) -> None:
        self.ticket = ticket

    def fetch(self, own_device):
        return self.ticket.result()[0][0], None


class _Pending:
    """The embedding of a node whose device work is queued on an ``ahead.Ahead``:
    ``fetch(own_device)`` -> ``(maps, stats)`` (a job nobody started yet runs on ``own_device``).
    A job that failed on the worker's context is run once more on the walk's own: two nodes in
    flight need more device memory than one, and what fails for that reason must not fail the
    run (a failure of the node itself just repeats)."""

    __slots__ = ("queue", "job", "again")

    def __init__(self, queue, job, again) -> None:
        self.queue, self.job, self.again = queue, job, again

    def fetch(self, own_device):
        try:
            return self.queue.result(self.job, own_device)
        except RuntimeError:
            return self.again(own_device)


def _small_path() -> bool:
    """SCS_NO_SMALL_PATH=1 (diagnostic) sends small nodes through the general per-node path
    (tables upload, build, contract, solve) instead of ``scs_small_solve``."""
    import os

    return not int(_env.probe("SCS_NO_SMALL_PATH", "0"))


def prepare_node(tables: fl.TreeTables, contract_edges: bool):
    """Contraction groups of a node and its tables renumbered so that every group is a
    consecutive id range: ``(work, perm, group_start or None, n_groups)``."""
    n = tables.n_taxa
    if contract_edges:
        groups = fl.contraction_groups(tables)
    else:
        groups = np.arange(n, dtype=np.int32)
    n_groups = int(groups.max()) + 1
    if n_groups < n:
        work, perm, group_start = relabel_for_contraction(tables, groups)
    else:
        work, perm, group_start = tables, np.arange(n, dtype=np.int32), None
    return work, perm, group_start, n_groups


def _labels_and_members(maps, random_state, n, perm, group_start, n_groups):
    """k_means on the embedding exactly as ``SpectralClustering.fit`` calls it
    (sklearn/cluster/_spectral.py:759-766) -- through ``kmeans2.labels``, which drives
    scikit-learn's compiled Lloyd iteration without the per-call validation around it when it
    has verified that it reproduces ``k_means`` bit for bit -- and the member taxa of every
    vertex."""
    from . import kmeans2

    if _node_trace is not None:
        global _traced_maps
        _traced_maps = np.array(maps, dtype=np.float64, copy=True)
    labels = kmeans2.labels(maps, random_state)
    if group_start is None:
        members = [np.array([i], dtype=np.int32) for i in range(n)]
    else:
        members = [perm[group_start[g] : group_start[g + 1]] for g in range(n_groups)]
    return members, np.asarray(labels)


def construct_supertree(
    trees: Sequence,
    weights: Sequence[float] | None = None,
    pcg_weighting: str = "one",
    *,
    contract_edges: bool = True,
    random_state: np.random.RandomState | None = None,
    team=None,
):
    """Spectral Cluster Supertree (SCS) -- see the reference docstring.

    Parameters and return value as in the reference
    (reference: src/sc_supertree/scs.py:18-59).

    Extension (keyword-only): ``team`` -- a ``partition.Team`` when several ranks walk the
    recursion together (one process per GPU; every rank passes the same trees and a
    RandomState in the same state).  Left at None a job launched by ``torch.distributed.run``
    finds its team from the environment, a plain run uses GPU ``SCS_DEVICE`` (default 0).
    """
    if team is None:
        team = default_team()
    if random_state is None:
        if team is not None and team.world > 1:
            # several ranks walk one recursion with ONE stream: rank 0 draws the seed for all
            # (an OS-seeded generator per rank would send the ranks down different recursions
            # and the next collective would meet mismatched shapes)
            mine = int(np.random.RandomState().randint(0, 2**31 - 1)) if team.rank == 0 else None
            random_state = np.random.RandomState(team.allgather(mine)[0])
        else:
            random_state = np.random.RandomState()

    if isinstance(trees, TreeArrays):
        # extension: a forest that was parsed straight into arrays (load.load_tree_arrays)
        if pcg_weighting not in ("one", "branch", "depth", "bootstrap"):
            msg = f"Invalid weighting strategy selected: '{pcg_weighting}'"
            raise ValueError(msg)
        arrays = trees
        if weights is not None:
            if len(weights) != arrays.n_trees:
                msg = (
                    f"The number of trees ({arrays.n_trees}) "
                    f"and tree weights ({len(weights)}) must match."
                )
                raise ValueError(msg)
            arrays = TreeArrays(arrays.n_taxa, arrays.node_off, arrays.parent, arrays.taxon, arrays.length,
                                arrays.support, np.asarray([float(w) for w in weights]), arrays.taxa)
        if arrays.n_trees == 0:
            msg = "There must be at least one tree to make a supertree."
            raise ValueError(msg)
        return _finish(_construct(arrays, pcg_weighting, contract_edges, random_state, team=team))

    if len(trees) == 0:
        msg = "There must be at least one tree to make a supertree."
        raise ValueError(msg)

    if pcg_weighting not in ("one", "branch", "depth", "bootstrap"):
        msg = f"Invalid weighting strategy selected: '{pcg_weighting}'"
        raise ValueError(msg)

    if weights is None:
        weights = [1.0 for _ in range(len(trees))]

    if len(trees) != len(weights):
        msg = (
            f"The number of trees ({len(trees)}) "
            f"and tree weights ({len(weights)}) must match."
        )
        raise ValueError(msg)

    pairs = [(t, w) for t, w in zip(trees, weights) if not is_not_completed(t)]
    if len(pairs) == 0:
        msg = "There must be at least one tree to make a supertree."
        raise ValueError(msg)
    trees = [t for t, _ in pairs]
    weights = [w for _, w in pairs]

    taxa = sorted(_all_tip_names(trees))
    arrays = TreeArrays.from_trees(trees, weights, taxa)
    return _finish(_construct(arrays, pcg_weighting, contract_edges, random_state, team=team))


def _finish(result):
    """A cogent3 tree when cogent3 is installed, as the reference returns."""
    if isinstance(result, TreeNode):
        try:
            import cogent3  # noqa: F401  # type: ignore[import-not-found]
        except ImportError:
            return result
        return _make_result_tree(result.get_newick())
    return result


def _all_tip_names(trees) -> set[str]:
    names: set[str] = set()
    for tree in trees:
        names.update(tree.get_tip_names())
    return names


def _induce(names: set[str], trees, weights):
    """reference: src/sc_supertree/scs.py:411-455"""
    out_trees, out_weights = [], []
    for tree, w in zip(trees, weights):
        if len(names.intersection(tree.get_tip_names())) < 2:
            continue
        sub = tree.get_sub_tree(names, ignore_missing=True, as_rooted=True)
        sub.name = "root"
        out_trees.append(sub)
        out_weights.append(w)
    return out_trees, out_weights


def _begin_small() -> bool:
    """Small children's solves are BEGUN (``scs_small_solve_begin``, a ticket) as soon as their tables are
    there and ended at the visit -- with or without the look-ahead worker (round 5: the tickets never
    needed it).  SCS_BEGIN_SMALL=0 (diagnostic): one waited-for batch per node when no worker runs."""
    import os

    return bool(int(_env.probe("SCS_BEGIN_SMALL", "1")))


def _ahead_enabled() -> bool:
    """SCS_AHEAD=0 (diagnostic) keeps all device work on the walk's own thread, node by node."""
    import os

    return bool(int(os.environ.get("SCS_AHEAD", "1") or 0))


def _construct(arrays: TreeArrays, pcg_weighting, contract_edges, random_state,
               bipartition=None, team=None, pre=None) -> TreeNode:
    """The recursion on flat tree arrays from its root node: ``_construct_node`` with, for a
    single-process run on the device path, a queue that lets the device work on nodes ahead of
    the walk (``ahead.Ahead``)."""
    if bipartition is None and getattr(arrays, "resident_device", None) is None:
        # the restriction step of the recursion runs on the device for forests big enough to be
        # worth a launch (treearrays.ResidentArrays, scs_forest_split); the device is only
        # created when such a split arrives (a run that never gets there touches none)
        arrays.resident_device = (lambda: team.solo) if team is not None else default_device
    if bipartition is None:
        from spectralclustersupertree_amd import levels

        levels.reset_stats()
    with _warm_allocator():
        return _construct_tuned(arrays, pcg_weighting, contract_edges, random_state, bipartition, team, pre)


_allocator_users = 0
_allocator_lock = threading.Lock()


class _warm_allocator:
    """glibc's mmap threshold raised for the duration of a recursion (``scs_host_malloc_tune``,
    csrc/scs_host.c: the node arrays of every split then come from the warm heap instead of fresh
    pages); counted, so that recursions on several threads (in-process teams) share it.
    OPT-IN since round 5 (``SCS_MALLOC_TUNE=1``): the forests of the recursion live on the device
    now (``treearrays.ResidentArrays``) and the host no longer allocates node arrays by the
    thousand, so a library call leaves the process-wide allocator settings alone by default."""

    def __enter__(self):
        import os

        global _allocator_users
        self.active = bool(int(os.environ.get("SCS_MALLOC_TUNE", "0") or 0))
        if not self.active:
            return self
        from spectralclustersupertree_amd import _hostlib

        with _allocator_lock:
            if _allocator_users == 0:
                _hostlib.load().scs_host_malloc_tune(1)
            _allocator_users += 1
        return self

    def __exit__(self, *exc):
        global _allocator_users
        if self.active:
            from spectralclustersupertree_amd import _hostlib

            with _allocator_lock:
                _allocator_users -= 1
                if _allocator_users == 0:
                    _hostlib.load().scs_host_malloc_tune(0)
        return False


_switch_users = 0
_switch_saved = None
_switch_lock = threading.Lock()


class _short_switch_interval:
    """The interpreter's switch interval lowered to 200 us while a recursion runs with its
    look-ahead worker (the worker comes back from a library call every few hundred microseconds
    and needs the interpreter for a few lines each time: it must not wait the default 5 ms for
    it).  Process-wide state: counted, so that recursions on several threads enter and leave in
    any order and the LAST one out restores what the FIRST one in found."""

    def __enter__(self):
        import sys

        global _switch_users, _switch_saved
        with _switch_lock:
            if _switch_users == 0:
                _switch_saved = sys.getswitchinterval()
                sys.setswitchinterval(min(_switch_saved, 2e-4))
            _switch_users += 1
        return self

    def __exit__(self, *exc):
        import sys

        global _switch_users, _switch_saved
        with _switch_lock:
            _switch_users -= 1
            if _switch_users == 0 and _switch_saved is not None:
                sys.setswitchinterval(_switch_saved)
                _switch_saved = None
        return False


def _engine_for(team) -> bool:
    """Whether a recursion walked by ``team`` goes through ``levels.Engine``: a single rank always; several ranks
    when they share the one stream (every rank then makes the same calls in the same order, and the engine deals the
    larger nodes of a level over them) and the team has not switched it off (``Team.level_engine``)."""
    if team is None or team.world == 1:
        return True
    return team.child_rng == "shared" and bool(getattr(team, "level_engine", True))


def _construct_tuned(arrays, pcg_weighting, contract_edges, random_state, bipartition, team, pre):
    # (a team in "shared" mode walks the same recursion on every rank: each rank has look-ahead workers on its own
    # GPU, and the level engine deals the larger nodes of a level over the ranks -- levels.Engine._process)
    single = _engine_for(team)
    if bipartition is None and single and _small_path() and _ahead_enabled():
        from spectralclustersupertree_amd.ahead import Ahead

        with _short_switch_interval():
            # (the worker's context is made when the first job arrives -- by then the walk has solved
            # the root on its own; a recursion that never reaches the spectral step touches no device)
            def second_context():
                return Device((team.solo if team is not None else default_device()).index)

            with Ahead(second_context, workers=int(os.environ.get("SCS_AHEAD_WORKERS", "3") or 1)) as queue:
                global _last_ahead_stats
                try:
                    return _construct_node(arrays, pcg_weighting, contract_edges, random_state, None, team, pre,
                                           queue)
                finally:
                    _last_ahead_stats = dict(queue.stats)
    return _construct_node(arrays, pcg_weighting, contract_edges, random_state, bipartition, team, pre, None)


def _construct_node(arrays: TreeArrays, pcg_weighting, contract_edges, random_state,
                    bipartition=None, team=None, pre=None, ahead=None) -> TreeNode:
    """One node of the recursion on flat tree arrays (reference: scs.py:96-174).

    Same decisions in the same order as the reference -- and therefore the same draws from
    ``random_state`` -- but the induced trees of the child problems come from ONE sweep of
    this node's forest (``TreeArrays.split``) instead of a ``get_sub_tree`` per tree and
    part on objects, and every child numbers its own taxa 0..k-1.

    Siblings are scheduled together: once a node's parts are known, the
    device work of every small single-component child (tables -> W -> contraction -> Jacobi ->
    embedding, ``scs_small_solve``) runs as ONE batched launch; it depends on no random
    draw, so each child later consumes the stream exactly where the reference does and only
    finds its embedding ready (``pre``).  With ``ahead`` the larger single-component children that
    are not next in the walk are queued for a second context on the same GPU: they are built
    and solved while the walk is busy with their left siblings' subtrees.

    ``team`` (several ranks walking together): nodes of at least ``team.shard_min`` vertices
    are solved collectively.  Below it, ``team.child_rng == "shared"`` has every rank walk
    every child with the shared stream (the reference's results, bit for bit): the subtrees go
    through ``levels.Engine`` on every rank, which DEALS the larger nodes of a level over the
    ranks -- an embedding depends on the node's forest alone, so whoever computes it, the labels
    drawn from it are the single-device ones (``Engine._process``); ``"forked"`` deals sibling
    sub-problems to the ranks, one per device, each with a RandomState forked from the parent's
    stream, and exchanges the subtrees (no label parity with a single-device run).
    """
    given = bipartition  # a caller's own routine (tests) is handed down unchanged
    name = arrays.name
    if arrays.n_trees == 1:  # reference: scs.py:96-98
        return arrays.to_tree(0)

    if pre is not None:
        present, tables, comp, presolved = pre
    else:
        present = arrays.present_taxa()
        if len(present) <= 2:
            return tip_names_to_tree([name(i) for i in present])
        # a node numbers its taxa by sorted name: ids are ranks of the sorted names
        tables = arrays.flatten(pcg_weighting, local_ids=present)
        comp = fl.pcg_components(tables)
        presolved = None
    if len(present) <= 2:
        return tip_names_to_tree([name(i) for i in present])
    n_comp = int(comp.max()) + 1

    if n_comp == 1:
        if given is not None:
            members, labels = given(tables, random_state, contract_edges=contract_edges)
        else:
            members, labels = spectral_bipartition_device(tables, random_state, contract_edges=contract_edges,
                                                          team=team, presolved=presolved)
        parts: list[list[int]] = [[], []]
        for ids, lab in zip(members, labels):
            parts[int(lab)].extend(int(present[int(i)]) for i in ids)
        if _node_trace is not None:
            _node_trace.append({
                "vertices": [tuple(name(present[int(i)]) for i in ids) for ids in members],
                "labels": np.asarray(labels).copy(), "maps": _traced_maps if given is None else None})
    else:
        parts = [[] for _ in range(n_comp)]
        for i, c in enumerate(comp):
            parts[int(c)].append(int(present[i]))

    return _construct_children(arrays, parts, pcg_weighting, contract_edges, random_state, given, team, ahead)


def _construct_children(arrays, parts, pcg_weighting, contract_edges, random_state, given=None, team=None,
                        ahead=None) -> TreeNode:
    """The second half of a recursion node (reference: scs.py:139-174): the forest restricted to every part
    (ids of ``arrays``), the children's subtrees in the order of ``parts``, joined under a new root."""
    name = arrays.name
    forked = (team is not None and team.world > 1 and team.child_rng == "forked")
    if given is None and not forked and _small_path() and _engine_for(team):
        # (round 6) all children as the first level of ONE level-synchronous engine, straight from this node's
        # forest: no download of the children's tables, their embeddings side by side (levels.construct_parts)
        from spectralclustersupertree_amd import levels

        if levels.parts_wanted(arrays, parts):
            tree = levels.construct_parts(arrays, parts, pcg_weighting, contract_edges, random_state, team, ahead)
            if tree is not None:
                return tree
    # ---- the children: ONE sweep of this node's forest restricts it to every part (host),
    # then one batched launch for the small ones
    children: list = []  # ("tips", ids) | ["sub", ids, sub, pre]
    to_split = []
    for component in parts:
        if len(component) == 0:
            continue
        component = sorted(component)
        if len(component) <= 2:
            children.append(("tips", component))
            continue
        children.append(["sub", component, None, None])
        to_split.append(np.asarray(component, dtype=np.int32))
    subs = iter(arrays.split(to_split, strategy=pcg_weighting))
    for child in children:
        if child[0] != "sub":
            continue
        child[2] = next(subs)  # taxa renumbered 0 .. k-1 in the order of child[1]
        if child[2].n_trees == 0:
            # no source tree keeps two of these taxa: the reference's recursive call receives
            # an empty list and raises (reference: scs.py:63-65 reached from :158)
            msg = "There must be at least one tree to make a supertree."
            raise ValueError(msg)
    speculate = given is None and not forked and _small_path() and _engine_for(team)
    if speculate:
        # (round 6) a child of at most SCS_SPEC_MAX_TAXA taxa: its whole subtree level by level with provisional
        # labels (levels.Engine), verified against the true draws when the walk gets there
        from spectralclustersupertree_amd import levels

        for child in children:
            if child[0] == "sub" and levels.wanted(child[2], len(child[1])):
                child[3] = levels.SpecRoot()
    if given is None and not forked and _small_path():
        _presolve_small_children(children, pcg_weighting, contract_edges, team, ahead)

    child_trees: list = []
    dealt: list[tuple[int, int, TreeArrays, np.random.RandomState]] = []  # (slot, owner, sub, rng)
    for child in children:
        if child[0] == "tips":
            child_trees.append(tip_names_to_tree([name(i) for i in child[1]]))
            continue
        _, component, sub, child_pre = child
        if forked and len(component) < team.shard_min:
            # one sub-problem per device: its own stream, forked here, in order, on every rank
            rng = np.random.RandomState(random_state.randint(0, 2**31 - 1))
            dealt.append((len(child_trees), len(dealt) % team.world, sub, rng))
            child_trees.append(None)
        elif speculate and isinstance(child_pre, levels.SpecRoot):
            child_trees.append(levels.construct(sub, pcg_weighting, contract_edges, random_state, team, ahead))
        else:
            child_trees.append(_construct_node(sub, pcg_weighting, contract_edges, random_state, given, team,
                                               child_pre, ahead))
        if len(sub.present_taxa()) < len(component):  # taxa no surviving tree holds (scs.py:166-170)
            covered = set(int(i) for i in sub.present_taxa())  # (ids of the child: positions in component)
            child_trees.extend(TreeNode(name(x)) for j, x in enumerate(component) if j not in covered)
    if dealt:
        from spectralclustersupertree_amd.partition import Team

        alone = Team(rank=0, world=1, device=team.solo, solo=team.solo)
        # every rank solves its share, then the subtrees are exchanged in FLAT form (lists:
        # pickling linked nodes recurses once per tree level); a failure on one rank travels
        # with the exchange, so that all ranks raise instead of waiting for one another
        mine, failure = {}, None
        try:
            for slot, owner, sub, rng in dealt:
                if owner == team.rank:
                    mine[slot] = _construct_node(sub, pcg_weighting, contract_edges, rng, given, alone).to_flat()
        except Exception as exc:  # noqa: BLE001 - re-raised on every rank below
            failure = f"rank {team.rank}: {type(exc).__name__}: {exc}"
        gathered = team.allgather((mine, failure))
        errors = [f for _, f in gathered if f is not None]
        if errors:
            msg = "a dealt sub-problem failed -- " + "; ".join(errors)
            raise RuntimeError(msg)
        for part, _ in gathered:
            for slot, flat in part.items():
                child_trees[slot] = TreeNode.from_flat(flat)
    return connect_trees(child_trees)


def _presolve_small_children(children, pcg_weighting, contract_edges, team, ahead=None) -> None:
    """Flatten every child problem, and run the device work of those that are one component of
    at most 128 taxa as ONE ``scs_small_solve`` batch; fills the ``pre`` slot of each child
    (present taxa, tables, components, embedding-or-None).  With ``ahead`` every small child's
    solve is BEGUN as soon as its tables are flattened (``scs_small_solve_begin``: the next child
    is flattened meanwhile, the result is fetched at the visit), and every larger single-component
    child becomes a job of the queue (``_Begun`` / ``_Pending`` in the ``pre`` slot)."""
    batch, where = [], []
    small_dev = None
    for child in children:
        if child[0] != "sub" or child[3] is not None:  # (not None: a speculative subtree root, levels.SpecRoot)
            continue
        sub = child[2]
        if sub.n_trees == 1:
            continue  # grafted as it is (reference: scs.py:96-98)
        present = sub.present_taxa()
        if len(present) <= 2:
            child[3] = (present, None, None, None)
            continue
        tables = sub.flatten(pcg_weighting, local_ids=present)
        comp = fl.pcg_components(tables)
        child[3] = [present, tables, comp, None]
        if int(comp.max()) != 0:
            continue
        if tables.n_taxa <= Device.SMALL_MAX_TAXA:
            work, perm, group_start, n_groups = prepare_node(tables, contract_edges)
            if ahead is not None or _begin_small():
                # begun at once, not waited for: the next child is flattened meanwhile, and a right
                # sibling's embedding is long there when the walk arrives
                if small_dev is None:
                    small_dev = team.solo if team is not None else default_device()
                ticket = small_dev.small_solve_begin([(work, group_start)])
                child[3][3] = (work, perm, group_start, n_groups, _Begun(ticket))
            else:
                batch.append((work, group_start))
                where.append((child, work, perm, group_start, n_groups))
        elif ahead is not None:
            # queued at once, the child the walk enters next included: its solve starts while the
            # remaining children are still being flattened
            work, perm, group_start, n_groups = prepare_node(tables, contract_edges)
            solve = lambda dev, work=work, group_start=group_start: _solve_node(dev, work, group_start)  # noqa: E731
            child[3][3] = (work, perm, group_start, n_groups, _Pending(ahead, ahead.submit(solve), solve))
    if batch:  # (a recursion that never reaches the spectral step never touches the device)
        dev = team.solo if team is not None else default_device()
        for (child, work, perm, group_start, n_groups), (maps, _) in zip(where, dev.small_solve(batch)):
            child[3][3] = (work, perm, group_start, n_groups, maps)
    for child in children:
        if child[0] == "sub" and isinstance(child[3], list):
            child[3] = tuple(child[3])


def _construct_objects(trees, weights, pcg_weighting, contract_edges, random_state,
                       bipartition=None) -> TreeNode:
    """The reference's recursion on tree objects (reference: scs.py:96-174); kept for the
    equivalence tests of the array path."""
    if bipartition is None:
        bipartition = spectral_bipartition_device
    if len(trees) == 1:  # reference: scs.py:96-98
        only = trees[0]
        newick = only.get_newick()
        from spectralclustersupertree_amd.tree import make_tree

        copy = make_tree(newick)
        for node in copy.iter_nontips(include_self=True):
            node.name = ""
        return copy

    all_names = _all_tip_names(trees)
    if len(all_names) <= 2:
        return tip_names_to_tree(sorted(all_names))

    taxa = sorted(all_names)
    tables = fl.flatten_trees(trees, weights, pcg_weighting, taxa)
    comp = fl.pcg_components(tables)
    n_comp = int(comp.max()) + 1

    if n_comp == 1:
        members, labels = bipartition(tables, random_state, contract_edges=contract_edges)
        parts: list[set[str]] = [set(), set()]
        for ids, lab in zip(members, labels):
            parts[int(lab)].update(taxa[int(i)] for i in ids)
    else:
        parts = [set() for _ in range(n_comp)]
        for i, c in enumerate(comp):
            parts[int(c)].add(taxa[i])

    child_trees: list[TreeNode] = []
    for component in parts:
        if len(component) == 0:
            continue
        if len(component) <= 2:
            child_trees.append(tip_names_to_tree(sorted(component)))
            continue
        sub_trees, sub_weights = _induce(component, trees, weights)
        if len(sub_trees) == 0:  # reference: scs.py:63-65 reached from :158
            msg = "There must be at least one tree to make a supertree."
            raise ValueError(msg)
        child_trees.append(
            _construct_objects(sub_trees, sub_weights, pcg_weighting, contract_edges, random_state,
                               bipartition)
        )
        missing = component.difference(_all_tip_names(sub_trees))
        child_trees.extend(TreeNode(x) for x in sorted(missing))
    return connect_trees(child_trees)
