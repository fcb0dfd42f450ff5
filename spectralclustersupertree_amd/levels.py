"""A whole subtree of the recursion level by level (round 6).

The reference walks ``construct_supertree`` depth-first, one node at a time, threading ONE
``RandomState`` through every node (reference: src/sc_supertree/scs.py:139-171): a node's labels need
its embedding, its children need its labels, and the labels are drawn from a stream whose position
depends on everything the walk has visited before.  On the device that order made the deep recursion
a chain of tens of thousands of sub-millisecond round trips.  Two facts break the chain:

* a node's EMBEDDING depends on its forest alone (the ARPACK start vector is drawn and not used,
  ``scs.spectral_bipartition_device``), and
* the PARTITION k-means finds on that embedding hardly ever depends on its draws (ten k-means++ starts
  of a two-cluster problem on what is in effect one coordinate) -- only which part is NUMBERED 0 does,
  i.e. the order in which the two children are visited.

So a subtree is first solved **level by level with provisional labels** (``Engine``): every level of the
subtree is ONE forest on the device (the trees of its nodes one after the other, the nodes' taxa as
consecutive ranges of one numbering), ONE ``scs_forest_split_level`` restricts all nodes of a level to all
of their parts and reports -- from the device -- present taxa, connected components and contraction
signatures of every child, ONE ``scs_small_solve_begin_level`` embeds all small nodes of the level, and the
host assigns labels from a private stream.  Then the walk proper (``Engine.build``) visits the nodes in the
reference's order with the caller's ``RandomState``: it draws what the reference draws (the start vector,
then scikit-learn's k-means on the stored embedding), and where the true partition equals the provisional
one -- labels equal or swapped -- the children are already there.  Where it does not (a tie that the draws
decide), that node's forest comes back from the device and its subtree is redone on the node-by-node
path with the true labels.  **Labels always come from the true draws**: the result is the reference's,
bit for bit, whatever the provisional labels were.

**Several ranks** (a ``partition.Team`` with the shared stream, one process per GPU): every rank walks the same
engine -- the level splits, the small batches and the label assignments are cheap and done by all -- and the
level's LARGER nodes, where the device time is, are dealt (``deal``): a node of ``team.shard_min`` vertices and more
is solved by all ranks together on their job-wide contexts (row-partitioned W, all-gather of the Krylov block), the
others go longest-first onto the least loaded rank, and one ``Team.allgather`` per level hands every embedding to
every rank.  The embedding being a function of the forest alone, the labels drawn from it are the single device's
whoever computed it: parity with a one-GPU run, which the "forked" streams of rounds 2-5 gave up.
"""

from __future__ import annotations

import os
import threading
import time

import numpy as np

from spectralclustersupertree_amd import _env
from spectralclustersupertree_amd import _native as nv
from spectralclustersupertree_amd import flatten as fl
from spectralclustersupertree_amd.tree import TreeNode, connect_trees, tip_names_to_tree
from spectralclustersupertree_amd.treearrays import _STRATEGY_CODE, ResidentArrays, TreeArrays

TIPS, GRAFT, COMPS, SPECTRAL, EMPTY, FALLBACK = 0, 1, 2, 3, 4, 5
MAX_PARTS = 8  # scs_forest_split_level: one mark bit per part

# diagnostics of the latest recursion (tests, tools/full_recursion_check.py)
stats = {"roots": 0, "from_parts": 0, "levels": 0, "nodes": 0, "spectral": 0, "mismatches": 0, "mismatch_sizes": [], "fallbacks": 0,
         "exact_group_nodes": 0, "t_first": 0.0, "t_host": 0.0, "t_small": 0.0, "t_large": 0.0, "t_labels": 0.0,
         "t_split": 0.0, "t_build": 0.0, "t_redo": 0.0, "n_large": 0, "deferred": 0, "deferred_agree": 0, "big_jobs": [], "redo_log": [], "split_log": [],
         "team_dealt": 0, "team_received": 0, "team_collective": 0, "lazy_nodes": 0, "t_lazy_wait": 0.0}


def reset_stats() -> None:
    stats.update({"roots": 0, "from_parts": 0, "levels": 0, "nodes": 0, "spectral": 0, "mismatches": 0, "mismatch_sizes": [],
                  "fallbacks": 0, "exact_group_nodes": 0, "t_first": 0.0, "t_host": 0.0, "t_small": 0.0, "t_large": 0.0,
                  "t_labels": 0.0, "t_split": 0.0, "t_build": 0.0, "t_redo": 0.0, "n_large": 0, "deferred": 0, "deferred_agree": 0, "big_jobs": [], "redo_log": [], "split_log": [],
                  "team_dealt": 0, "team_received": 0, "team_collective": 0, "lazy_nodes": 0, "t_lazy_wait": 0.0})


def max_taxa() -> int:
    """Largest node whose subtree is solved level by level (SCS_SPEC_MAX_TAXA; 0 switches the engine off)."""
    return int(os.environ.get("SCS_SPEC_MAX_TAXA", "2048") or 0)


def min_nodes() -> int:
    """Smallest forest (tree nodes) worth the engine's launches (SCS_SPEC_MIN_NODES)."""
    return int(_env.probe("SCS_SPEC_MIN_NODES", "4000"))


class SpecRoot:
    """Marker in a child's ``pre`` slot: this child's subtree goes through ``construct``."""


def wanted(sub, n_component: int) -> bool:
    """Whether the subtree of child ``sub`` (``n_component`` taxa) goes through the engine.  Any size does (since
    the second half of round 6): nodes of more than ``max_taxa()`` vertices are embedded by the engine like all
    others -- the level's split without a download, the analysis on the device, both children of a node side by side
    on the look-ahead workers -- but nothing is computed below them on a guess (``Engine._provisional``: deferred)."""
    cap = max_taxa()
    if cap <= 0 or n_component <= 2 or sub.n_trees < 2:
        return False
    if n_component > cap and not int(_env.probe("SCS_SPEC_ABOVE_CAP", "1")):
        return False  # (diagnostic: nodes above the cap on the node-by-node path, as in the first half of round 6)
    n_nodes = sub.n_nodes if isinstance(sub, ResidentArrays) else len(sub.parent)
    return n_nodes >= min_nodes()


def construct(sub, pcg_weighting, contract_edges, random_state, team=None, ahead=None) -> TreeNode:
    """The subtree of ``sub`` (a child forest of a split): ``scs._construct_node``'s result, same draws."""
    from spectralclustersupertree_amd import scs

    present = sub.present_taxa()
    if sub.n_trees == 1 or len(present) <= 2:
        return scs._construct_node(sub, pcg_weighting, contract_edges, random_state, None, team, None, ahead)
    dev = team.solo if team is not None else scs.default_device()
    engine = Engine(sub, pcg_weighting, contract_edges, dev, team, ahead)
    try:
        engine.run()
    except nv.ScsError as exc:
        # (several ranks: a rank that left the engine alone would leave the others waiting in an exchange)
        if exc.code != nv.ENOMEM or engine._spread():
            raise
        # not enough device memory for the levels of this subtree beside what else is resident: node by node
        engine.levels.clear()
        return scs._construct_node(sub, pcg_weighting, contract_edges, random_state, None, team, None, ahead)
    t0 = time.perf_counter()
    try:
        return engine.build(0, 0, random_state)
    finally:
        stats["t_build"] += time.perf_counter() - t0  # (nested engines of redone subtrees count twice)


def parts_wanted(arrays, parts) -> bool:
    """Whether the children of a node of the node-by-node walk (its forest ``arrays``, its ``parts``) are taken by ONE
    engine from the node's own forest (``construct_parts``)."""
    if max_taxa() <= 0 or arrays.n_trees < 1 or _env.probe("SCS_SPEC_FROM_PARTS", "1") == "0":
        return False
    splitting = sum(1 for p in parts if len(p) > 2)
    if splitting < 1 or splitting > MAX_PARTS:
        return False
    n_nodes = arrays.n_nodes if isinstance(arrays, ResidentArrays) else len(arrays.parent)
    return n_nodes >= min_nodes()


def construct_parts(arrays, parts, pcg_weighting, contract_edges, random_state, team=None, ahead=None) -> TreeNode:
    """The second half of a node of the node-by-node walk (``scs._construct_children``: the forest restricted to
    every part, the children's subtrees in the order of ``parts``, joined under a new root; reference:
    scs.py:139-174) -- with ONE level split of the node's own forest instead of ``scs_forest_split`` + a download
    of the children's tables, and all children as the first level of one engine: their embeddings run side by
    side, their components and contraction signatures come from the device."""
    from spectralclustersupertree_amd import scs

    dev = team.solo if team is not None else scs.default_device()
    engine = Engine(arrays, pcg_weighting, contract_edges, dev, team, ahead)
    try:
        first = engine.run_parts(parts)
    except nv.ScsError as exc:
        if exc.code != nv.ENOMEM or engine._spread():
            raise
        engine.levels.clear()
        return None  # (the caller goes on node by node)
    t0 = time.perf_counter()
    try:
        return engine._children(0, list(range(int(first.node_seg[0]), int(first.node_seg[1]))), random_state)
    finally:
        stats["t_build"] += time.perf_counter() - t0


def deal(large, n_pres, n_groups, n_trees, gs_patch, world: int, shard_min: int):
    """The larger nodes of one level over the ranks of a team: ``(collective, owner)``.

    ``collective``: ``[(node, splits)]`` -- nodes of ``shard_min`` vertices and more that split into group-aligned row
    ranges for all ranks (``partition.row_splits``; a node that does not is dealt like the others), largest first: the
    ONE order in which every rank enters them.  ``owner[node]``: the rank that embeds a dealt node -- longest job
    first onto the least loaded rank (cost: vertices squared times trees, what the tile kernel does), ties to the lower
    rank.  A pure function of its arguments: every rank computes the same deal."""
    from spectralclustersupertree_amd.partition import row_splits

    order = sorted((int(k) for k in large), key=lambda k: (-int(n_groups[k]), k))
    collective, owner = [], {}
    load = [0.0] * world
    for k in order:
        if int(n_groups[k]) >= shard_min:
            try:
                collective.append((k, row_splits(int(n_pres[k]), world, gs_patch.get(k))))
                continue
            except ValueError:
                pass  # fewer groups (or 256-row blocks) than ranks
        r = min(range(world), key=lambda r: (load[r], r))
        owner[k] = r
        load[r] += float(int(n_groups[k])) ** 2 * float(max(int(n_trees[k]), 1))
    return collective, owner


class _Lazy:
    """The embeddings of a level's nodes ABOVE the cap (``max_taxa()``: nothing is computed below them on a guess, so
    the engine does not need them -- only the walk does, when it gets there), as ONE job of the look-ahead queue that
    takes them one after the other in the order of the visit.

    Before, the two children of a node above the cap were built side by side and the level waited for both: the
    first one visited -- the only one the walk needs now -- finished when the pair did.  Now it has the chip to
    itself, the walk goes on into its subtree as soon as it is there, and its sibling's build runs behind that
    subtree's levels (whose small kernels leave most of the chip idle).  Same embeddings, same labels: only WHEN a
    node is embedded changes."""

    def __init__(self, engine, lev, nodes, relabel, gs_patch) -> None:
        self.order = [int(k) for k in nodes]
        self.fns = {k: engine._large_job(lev, k, relabel, gs_patch) for k in self.order}
        self.done = {k: threading.Event() for k in self.order}
        self.unfetched = set(self.order)
        self.value: dict = {}
        self.error: dict = {}
        self.claimed: set = set()
        self.lock = threading.Lock()
        self.job = engine.ahead.submit(self._run)

    def _compute(self, k: int, dev) -> None:
        with self.lock:
            if k in self.claimed:
                return
            self.claimed.add(k)
        try:
            self.value[k] = self.fns[k](dev)
        except BaseException as exc:  # noqa: BLE001 -- handed to the walk when it asks
            self.error[k] = exc
        finally:
            self.done[k].set()

    def _run(self, dev) -> None:
        for k in self.order:
            self._compute(k, dev)

    def pending(self, k: int) -> bool:
        return k in self.unfetched

    def fetch(self, k: int, own_device) -> np.ndarray:
        """The embedding of node ``k`` (once).  A sequence no worker has picked up yet: this node here, now."""
        from spectralclustersupertree_amd.ahead import _QUEUED

        self.unfetched.discard(k)
        done = self.done[k]
        if not done.is_set() and self.job.state == _QUEUED:
            self._compute(k, own_device)
        done.wait()
        fn = self.fns.pop(k)
        exc = self.error.pop(k, None)
        if exc is None:
            return self.value.pop(k)
        if isinstance(exc, RuntimeError):
            # what failed beside other work on a worker's context is solved once more on the walk's own (two nodes in
            # flight need more device memory than one; a failure of the node itself just repeats)
            return fn(own_device)
        raise exc


class Level:
    """One level of a speculative subtree: K nodes as consecutive tree ranges of ``forest`` and consecutive
    id ranges of its taxon numbering (struct of arrays; see ``Engine``)."""

    __slots__ = ("forest", "K", "T", "t_lo", "t_hi", "n_leaves", "u_lo", "u_sz", "gid", "present", "comp_root",
                 "sig", "kind", "n_pres", "v_off", "n_groups", "maps", "prov", "members", "sorted_taxa", "seg_start",
                 "seg_len", "seg_child", "node_seg", "graft", "monotone", "shift", "defer", "lazy")


class Engine:
    def __init__(self, sub, strategy: str, contract_edges: bool, dev, team=None, ahead=None) -> None:
        self.sub, self.strategy, self.contract, self.dev = sub, strategy, bool(contract_edges), dev
        self.team, self.ahead = team, ahead
        self.levels: list[Level] = []
        self.prov_rs = np.random.RandomState(0x5C5)  # the provisional labels' private stream
        self.weights_ok = bool(np.all(np.asarray(sub.weights) >= 0))
        self.small_max = dev.SMALL_MAX_TAXA

    # ------------------------------------------------------------------ level 0
    def _first_level(self) -> Level:
        sub, dev = self.sub, self.dev
        k = int(sub.n_taxa)
        lev = Level()
        if (isinstance(sub, ResidentArrays) and sub._tables is not None and sub.strategy == self.strategy
                and sub.forest.dev is dev and sub.forest._h):
            forest = sub.forest
            comp_root, sig = forest.analyze()
            present = np.zeros(k, dtype=np.uint8)
            present[sub.present_taxa()] = 1
            lev.monotone = bool(forest.monotone_flag)
        else:
            host = sub.to_host() if isinstance(sub, ResidentArrays) else sub
            res = ResidentArrays.from_host(host, dev)
            # the trees are a split's children (nothing to splice): restricting them to ALL their taxa
            # returns them as they are -- with their tables on the device, and the analysis of those
            forest, child_trees, _, present, comp_root, sig = res.forest.split_level(
                np.zeros(k, dtype=np.int32), np.arange(k, dtype=np.int32), 1, k, _STRATEGY_CODE[self.strategy],
                np.asarray([res.forest.n_trees], dtype=np.int32))
            if int(child_trees[0, 0]) != sub.n_trees:
                msg = "levels: a child forest changed under the identity restriction"
                raise AssertionError(msg)
            lev.monotone = bool(forest.monotone_flag)
        lev.forest = forest
        lev.shift = 0
        lev.K, lev.T = 1, k
        lev.t_lo = np.zeros(1, dtype=np.int32)
        lev.t_hi = np.asarray([forest.n_trees], dtype=np.int32)
        lev.n_leaves = np.asarray([forest.n_leaves], dtype=np.int64)
        lev.u_lo = np.zeros(1, dtype=np.int32)
        lev.u_sz = np.asarray([k], dtype=np.int32)
        lev.gid = np.arange(k, dtype=np.int32)
        lev.present, lev.comp_root, lev.sig = present, comp_root[:k], sig[:k]
        return lev

    # ------------------------------------------------------------------ all levels
    def run(self) -> None:
        stats["roots"] += 1
        t0 = time.perf_counter()
        lev = self._first_level()
        stats["t_first"] += time.perf_counter() - t0
        while lev is not None:
            self.levels.append(lev)
            stats["levels"] += 1
            stats["nodes"] += lev.K
            lev = self._process(lev)

    def run_parts(self, parts) -> Level:
        """The engine below a node whose parts are KNOWN (taxon ids of ``self.sub``): a first level that is that node
        alone, split into ``parts`` (only its node arrays are needed on the device), then level by level."""
        stats["roots"] += 1
        stats["from_parts"] += 1
        t0 = time.perf_counter()
        sub, dev = self.sub, self.dev
        if isinstance(sub, ResidentArrays) and sub.forest.dev is dev and sub.forest._h:
            forest = sub.forest
        else:
            host = sub.to_host() if isinstance(sub, ResidentArrays) else sub
            forest = ResidentArrays.from_host(host, dev).forest
        k = int(sub.n_taxa)
        l0 = Level()
        l0.forest, l0.shift, l0.monotone = forest, 0, True
        l0.K, l0.T = 1, k
        l0.t_lo = np.zeros(1, dtype=np.int32)
        l0.t_hi = np.asarray([forest.n_trees], dtype=np.int32)
        l0.n_leaves = np.zeros(1, dtype=np.int64)
        l0.u_lo = np.zeros(1, dtype=np.int32)
        l0.u_sz = np.asarray([k], dtype=np.int32)
        l0.gid = np.arange(k, dtype=np.int32)
        l0.present = np.zeros(k, dtype=np.uint8)
        l0.comp_root = l0.sig = None
        l0.kind = np.asarray([COMPS], dtype=np.int8)
        l0.n_pres = np.zeros(1, dtype=np.int64)
        l0.n_groups = np.zeros(1, dtype=np.int32)
        l0.v_off = np.zeros(2, dtype=np.int64)
        l0.maps = np.zeros((0, 2))
        l0.prov = np.zeros(0, dtype=np.int8)
        l0.graft, l0.members = {}, {}
        l0.defer = np.zeros(1, dtype=bool)
        tpart = np.full(k, -1, dtype=np.int64)
        for b, ids in enumerate(parts):
            if len(ids):
                ids = np.fromiter(ids, dtype=np.int64, count=len(ids))  # (a list or a set of ids)
                tpart[ids] = b
                l0.present[ids] = 1
        stats["t_first"] += time.perf_counter() - t0
        stats["levels"] += 1
        self.levels.append(l0)
        nxt = self._split(l0, tpart, np.zeros(k, dtype=np.int32))
        while nxt is not None:
            self.levels.append(nxt)
            stats["levels"] += 1
            stats["nodes"] += nxt.K
            nxt = self._process(nxt)
        return l0

    def _monotone(self, lev: Level) -> bool:
        return self.strategy in ("one", "depth", "branch") and lev.monotone and self.weights_ok

    def _process(self, lev: Level):
        """Classify the nodes of ``lev``, embed its spectral nodes, label them provisionally, split the
        level's forest into the next level (None when no node has a part left to split)."""
        t_begin = time.perf_counter()
        t_dev = 0.0
        K, T = lev.K, lev.T
        u_lo, u_sz = lev.u_lo, lev.u_sz
        nid = np.repeat(np.arange(K, dtype=np.int32), u_sz)
        pres = lev.present.astype(bool)
        m = (lev.t_hi - lev.t_lo).astype(np.int64)
        pres_i = pres.astype(np.int32)
        n_pres = np.add.reduceat(pres_i, u_lo)
        cs_pres = np.concatenate(([0], np.cumsum(pres_i)))  # exclusive prefix counts
        prank = (cs_pres[:-1] - cs_pres[u_lo][nid]).astype(np.int32)  # rank among the node's present taxa
        is_root = (lev.comp_root == np.arange(T, dtype=np.int32)) & pres
        root_i = is_root.astype(np.int32)
        n_comp = np.add.reduceat(root_i, u_lo)
        kind = np.full(K, SPECTRAL, dtype=np.int8)
        kind[n_comp > 1] = COMPS
        kind[n_pres <= 2] = TIPS
        kind[m == 1] = GRAFT
        kind[m == 0] = EMPTY
        lev.kind, lev.n_pres = kind, n_pres
        lev.graft, lev.members = {}, {}
        for k in np.flatnonzero(kind == GRAFT):
            lev.graft[int(k)] = self._graft(lev, int(k))

        # ---- spectral nodes: contraction groups (exact routine only where signatures collide)
        spectral = kind == SPECTRAL
        vertex_of = prank.copy()  # vertex of a present taxon (its group after contraction)
        relabel = np.where(pres, prank, 0).astype(np.int32)  # id in the node's (contracted) numbering
        n_groups = np.where(spectral, n_pres, 0).astype(np.int32)
        gs_patch: dict[int, np.ndarray] = {}
        if self.contract and spectral.any():
            idx = np.flatnonzero(pres & spectral[nid])
            if len(idx) > 1:
                s0, s1, nn = lev.sig[idx, 0], lev.sig[idx, 1], nid[idx]
                order = np.lexsort((s1, s0, nn))
                a, b, c = nn[order], s0[order], s1[order]
                eq = (a[1:] == a[:-1]) & (b[1:] == b[:-1]) & (c[1:] == c[:-1])
                for k in np.unique(a[1:][eq]):
                    self._exact_groups(lev, int(k), prank, vertex_of, relabel, n_groups, gs_patch)
        lev.n_groups = n_groups
        v_ptr = np.concatenate(([0], np.cumsum(n_groups, dtype=np.int64)))
        lev.v_off = v_ptr
        lev.maps = np.zeros((int(v_ptr[-1]), 2))
        lev.prov = np.zeros(int(v_ptr[-1]), dtype=np.int8)
        stats["spectral"] += int(spectral.sum())

        # ---- embeddings: the small nodes of the level in batches, the larger ones one by one
        small = np.flatnonzero(spectral & (n_pres <= self.small_max))
        t1 = time.perf_counter()
        large = np.flatnonzero(spectral & (n_pres > self.small_max))
        # several ranks walking this engine together (a team's shared stream): the level's larger nodes are DEALT --
        # an embedding depends on the node's forest alone, so every rank may take it from whoever computed it
        spread = self._spread() and len(large) > 0
        collective: list = []
        mine = [int(k) for k in large]
        if spread:
            collective, owner = deal(large, n_pres, n_groups, m, gs_patch, self.team.world, self.team.shard_min)
            mine = [k for k in mine if owner.get(k) == self.team.rank]
        # nodes above the cap are the walk's to label and nothing is computed below them on a guess: the level does
        # not wait for them (``_Lazy``; a team's ranks enter them together instead, above)
        lev.lazy = None
        lazy_nodes: list[int] = []
        if self.ahead is not None and not spread and max_taxa() > 0 and _env.probe("SCS_SPEC_LAZY", "1") != "0":
            lazy_nodes = [k for k in mine if int(n_groups[k]) > max_taxa()]
            if lazy_nodes:
                mine = [k for k in mine if k not in lazy_nodes]
                lev.lazy = _Lazy(self, lev, lazy_nodes, relabel, gs_patch)
                stats["lazy_nodes"] += len(lazy_nodes)
        got: dict[int, np.ndarray] = {}
        failure = None
        jobs = []
        try:
            if self.ahead is not None and (len(mine) > 1 or (mine and (collective or len(small)) and spread)):
                # the larger nodes of the level side by side on the look-ahead workers' contexts (largest first),
                # while this thread batches the small ones -- and then takes whatever nobody has started
                for k in sorted(mine, key=lambda k: -int(n_pres[k])):
                    jobs.append((k, self.ahead.submit(self._large_job(lev, k, relabel, gs_patch))))
        except Exception as exc:  # noqa: BLE001 - under a team the failure travels with the exchange below
            if not spread:
                raise
            failure = f"rank {self.team.rank}: {type(exc).__name__}: {exc}"
        if len(small):
            self._solve_small(lev, small, m, relabel, nid, gs_patch)
        t2 = time.perf_counter()
        # (nodes of team.shard_min vertices and more: every rank its rows, on the job-wide context, in ONE order)
        for k, splits in collective:
            got[k] = self._collective_job(lev, k, relabel, gs_patch, splits)
            stats["team_collective"] += 1
        try:
            if failure is not None:
                pass
            elif jobs:
                for k, job in jobs:
                    try:
                        got[k] = self.ahead.result(job, self.dev)
                    except RuntimeError:
                        # several nodes in flight need more device memory than one: what failed on a worker's context
                        # is solved once more on this thread's own (a failure of the node itself just repeats)
                        got[k] = self._large_job(lev, k, relabel, gs_patch)(self.dev)
            else:
                for k in mine:
                    got[k] = self._large_job(lev, k, relabel, gs_patch)(self.dev)
        except Exception as exc:  # noqa: BLE001
            if not spread:
                raise
            failure = f"rank {self.team.rank}: {type(exc).__name__}: {exc}"
        if spread:
            got = self._exchange(got, failure, collective, mine)
        for k, maps_k in got.items():
            lev.maps[int(v_ptr[k]):int(v_ptr[k + 1])] = maps_k
        stats["n_large"] += len(large)
        t3 = time.perf_counter()
        # ---- provisional labels
        from spectralclustersupertree_amd import kmeans2

        # (a node whose batched solve failed has left the kind: its rows of maps are zero, its labels unused)
        prov, defer = self._provisional(lev, v_ptr, lazy_nodes)
        lev.defer = defer if defer is not None else np.diff(v_ptr) > max_taxa()
        stats["deferred"] += int(lev.defer.sum())
        if prov is not None:
            lev.prov[:] = prov
        else:
            for k in np.flatnonzero(kind == SPECTRAL):
                if int(k) in lazy_nodes:
                    continue
                v0, v1 = int(v_ptr[k]), int(v_ptr[k + 1])
                lev.prov[v0:v1] = kmeans2.labels(lev.maps[v0:v1], self.prov_rs)
        t4 = time.perf_counter()
        stats["t_small"] += t2 - t1
        stats["t_large"] += t3 - t2
        stats["t_labels"] += t4 - t3
        t_dev = t4 - t1

        # ---- part of every taxon inside its node
        tpart = np.full(T, -1, dtype=np.int64)
        comps = kind == COMPS
        if comps.any():
            cs_root = np.concatenate(([0], np.cumsum(root_i)))
            sel = pres & comps[nid]
            tpart[sel] = cs_root[lev.comp_root[sel]] - cs_root[u_lo[nid[sel]]]
        spectral = (kind == SPECTRAL) & ~lev.defer  # (a deferred node has no provisional parts)
        if spectral.any():
            sel = pres & spectral[nid]
            tpart[sel] = lev.prov[v_ptr[nid[sel]] + vertex_of[sel]]
        stats["t_host"] += time.perf_counter() - t_begin - t_dev
        return self._split(lev, tpart, nid)

    def _split(self, lev: Level, tpart: np.ndarray, nid: np.ndarray):
        """The parts of every node (``tpart``: part of a taxon inside its node, -1 none) sorted by (node, part,
        taxon), and ONE ``scs_forest_split_level`` for all parts of more than two taxa: the next level."""
        t_begin = time.perf_counter()
        t_dev = 0.0
        K, T = lev.K, lev.T
        kind = lev.kind
        valid = np.flatnonzero(tpart >= 0)
        if len(valid) == 0:
            lev.sorted_taxa = np.zeros(0, dtype=np.int32)
            lev.seg_start = lev.seg_len = lev.seg_child = np.zeros(0, dtype=np.int64)
            lev.node_seg = np.zeros(K + 1, dtype=np.int64)
            stats["t_host"] += time.perf_counter() - t_begin - t_dev
            return None
        pmax = int(tpart.max()) + 1
        key = nid[valid].astype(np.int64) * pmax + tpart[valid]
        order = np.argsort(key, kind="stable")
        skey = key[order]
        lev.sorted_taxa = valid[order].astype(np.int32)
        first = np.concatenate(([True], skey[1:] != skey[:-1]))
        seg_start = np.flatnonzero(first)
        seg_len = np.diff(np.concatenate((seg_start, [len(skey)])))
        seg_node = (skey[seg_start] // pmax).astype(np.int64)
        lev.seg_start, lev.seg_len = seg_start, seg_len
        lev.node_seg = np.searchsorted(seg_node, np.arange(K + 1))  # segments of node k: [node_seg[k], node_seg[k + 1])
        splitting = seg_len > 2
        cs_split = np.concatenate(([0], np.cumsum(splitting.astype(np.int64))))
        slot = cs_split[:-1] - cs_split[lev.node_seg[seg_node]]  # index among the node's splitting parts
        too_many = np.unique(seg_node[splitting & (slot >= MAX_PARTS)])
        if len(too_many):
            # more than eight parts to restrict to (many components): that node goes node by node
            kind[too_many] = FALLBACK
            splitting &= ~np.isin(seg_node, too_many)
        lev.seg_child = np.full(len(seg_start), -1, dtype=np.int64)
        child_seg = np.flatnonzero(splitting)
        if len(child_seg) == 0:
            stats["t_host"] += time.perf_counter() - t_begin - t_dev
            return None
        # ---- the next level: children ordered (slot, node) -- the tree order of the union forest
        child_seg = child_seg[np.argsort(slot[child_seg] * K + seg_node[child_seg], kind="stable")]
        lev.seg_child[child_seg] = np.arange(len(child_seg))
        lens = seg_len[child_seg]
        starts_new = np.cumsum(lens) - lens
        total = int(lens.sum())
        pos = np.repeat(seg_start[child_seg] - starts_new, lens) + np.arange(total)
        taxa_new = lev.sorted_taxa[pos]
        part_of = np.full(T, -1, dtype=np.int32)
        new_id = np.zeros(T, dtype=np.int32)
        part_of[taxa_new] = np.repeat(slot[child_seg], lens).astype(np.int32)
        new_id[taxa_new] = np.arange(total, dtype=np.int32)
        n_parts = int(slot[child_seg].max()) + 1
        if lev.forest.n_taxa != T:
            # a slice of a larger level's forest (``redo``): its taxa keep the ids of that level
            wide = np.full(lev.forest.n_taxa, -1, dtype=np.int32)
            wide[lev.shift:lev.shift + T] = part_of
            part_of = wide
            wide = np.zeros(lev.forest.n_taxa, dtype=np.int32)
            wide[lev.shift:lev.shift + T] = new_id
            new_id = wide
        t5 = time.perf_counter()
        union, child_trees, child_leaves, present, comp_root, sig = lev.forest.split_level(
            part_of, new_id, n_parts, total, _STRATEGY_CODE[self.strategy], lev.t_hi)
        t6 = time.perf_counter()
        stats["t_split"] += t6 - t5
        if t6 - t5 >= 0.005:
            stats["split_log"].append((int(lev.n_leaves.sum()), round(t6 - t5, 4)))
        t_dev += t6 - t5
        nxt = Level()
        nxt.forest = union
        nxt.shift = 0
        nxt.monotone = bool(union.monotone_flag)
        nxt.K, nxt.T = len(child_seg), total
        cs, cn = slot[child_seg], seg_node[child_seg]
        counts = child_trees[cs, cn].astype(np.int64)
        t_hi = np.cumsum(counts)
        nxt.t_lo = (t_hi - counts).astype(np.int32)
        nxt.t_hi = t_hi.astype(np.int32)
        nxt.n_leaves = child_leaves[cs, cn].astype(np.int64)
        nxt.u_lo = starts_new.astype(np.int32)
        nxt.u_sz = lens.astype(np.int32)
        nxt.gid = lev.gid[taxa_new]
        nxt.present, nxt.comp_root, nxt.sig = present, comp_root, sig
        stats["t_host"] += time.perf_counter() - t_begin - t_dev
        return nxt

    def _provisional(self, lev: Level, v_ptr: np.ndarray, skip=()):
        """Provisional labels of all spectral nodes of the level, and which nodes to DEFER.

        Which partition the labels of record will be is a draw from a distribution (ten k-means++ starts, the best
        kept).  For nearly all nodes that distribution is one point; for a few -- a Fiedler vector without a gap
        between its two clusters -- it is close to a coin flip between two partitions, and everything computed
        below a wrong guess is thrown away (measured at configs[4]: 880 unconfirmed partitions of 61 436 cost a third
        of the engine's work, the largest at 2 019, 1 877, 1 608 vertices).  So the labels are assigned
        ``SCS_SPEC_VOTES`` times from different draws: a node on which the votes AGREE goes on with that partition;
        a node on which they do not is deferred -- nothing below it is computed now, the walk takes its subtree up
        again (``redo``) when it arrives there with the labels of record."""
        from spectralclustersupertree_amd import kmeans2

        # (measured at 20 000 taxa / 5 000 trees, profiles/r06_levels_votes.txt: 1 / 3 / 5 votes -> 19 510 / 14 200 /
        # 13 346 nodes computed for 12 400 needed, 178 / 76 / 55 bets lost -- and 6.1 / 6.7 / 6.9 s: what is no longer
        # computed in vain is computed later, one deferred subtree after the other, at the walk's pace.  Default: one.)
        votes = max(1, int(_env.probe("SCS_SPEC_VOTES", "1")))
        runs = []
        maps, ptr, rows = lev.maps, v_ptr, None
        if len(skip):
            # (nodes whose embedding is not there yet -- ``_Lazy``, all above the cap -- take no part: empty ranges)
            sizes = np.diff(v_ptr)
            rows = np.ones(len(maps), dtype=bool)
            for k in skip:
                rows[int(v_ptr[k]):int(v_ptr[k + 1])] = False
                sizes[k] = 0
            maps = maps[rows]
            ptr = np.concatenate(([0], np.cumsum(sizes, dtype=np.int64)))
        for _ in range(votes):
            lab = kmeans2.provisional_labels(maps, ptr, self.prov_rs)
            if lab is None:
                return None, None
            if rows is not None:
                full = np.zeros(len(rows), dtype=lab.dtype)
                full[rows] = lab
                lab = full
            runs.append(lab)
        sizes = np.diff(v_ptr)
        # a node above the cap is never guessed: a lost bet there throws away a subtree of thousands of taxa
        # (measured with a cap of 8 192: unconfirmed partitions at 6 145, 4 562, 3 810 vertices, 3.3 s of redone work
        # in a 8.6 s run) -- its labels are the walk's to assign
        defer = sizes > max_taxa()
        if votes == 1 or len(runs[0]) == 0:
            return runs[0], defer
        nodes = np.flatnonzero(sizes > 0)
        starts = v_ptr[nodes].astype(np.int64)
        seg = np.repeat(np.arange(len(nodes)), sizes[nodes])
        # the partition, whatever the numbering: every label relative to the node's first vertex
        canon = [lab ^ lab[starts][seg] for lab in runs]
        differ = np.zeros(len(nodes), dtype=bool)
        for i in range(1, votes):
            differ |= np.add.reduceat((canon[0] != canon[i]).astype(np.int32), starts) != 0
        min_defer = int(_env.probe("SCS_SPEC_DEFER_MIN", "8"))
        defer[nodes[differ & (sizes[nodes] >= min_defer)]] = True
        return runs[0], defer

    # ------------------------------------------------------------------ several ranks, one engine
    def _spread(self) -> bool:
        team = self.team
        return team is not None and team.world > 1 and team.allgather is not None

    def _exchange(self, got: dict, failure, collective, mine) -> dict:
        """Every rank's share of the level's larger nodes to every rank (``Team.allgather``: plain data).  A failure
        on one rank travels with the exchange, so that all ranks raise instead of waiting for one another."""
        own = {int(k): np.ascontiguousarray(got[k]) for k in mine if k in got}
        gathered = self.team.allgather(("levels.maps", own, failure))
        if any(not (isinstance(g, tuple) and len(g) == 3 and g[0] == "levels.maps") for g in gathered):
            msg = "levels: the ranks of the team are not walking the same recursion (an exchange out of step)"
            raise RuntimeError(msg)
        errors = [f for _, _, f in gathered if f is not None]
        if errors:
            msg = "a dealt node of the level failed -- " + "; ".join(errors)
            raise RuntimeError(msg)
        out = {int(k): got[k] for k, _ in collective}
        for r, (_, part, _) in enumerate(gathered):
            for k, maps_k in part.items():
                out[int(k)] = got[k] if r == self.team.rank else np.asarray(maps_k, dtype=np.float64)
            if r != self.team.rank:
                stats["team_received"] += len(part)
        stats["team_dealt"] += len(own)
        return out

    def _collective_job(self, lev, k, relabel, gs_patch, splits) -> np.ndarray:
        """A node of ``team.shard_min`` vertices and more, by ALL ranks of the team on their job-wide contexts: this
        rank's rows of W from the node's slice of the level's tables (each rank holds the level forest on its own
        GPU), group-aligned splits, the row-partitioned LOBPCG with its all-gather of the Krylov block
        (``scs.spectral_bipartition_device``'s sharded branch; reference: scs.py:210-258) -- returns the whole embedding."""
        from spectralclustersupertree_amd import scs

        team = self.team
        lo, sz = int(lev.u_lo[k]), int(lev.u_sz[k])
        rl = relabel[lo:lo + sz].copy()
        n, gs = int(lev.n_pres[k]), gs_patch.get(k)
        dtab = team.device.upload_range(lev.forest, int(lev.t_lo[k]), int(lev.t_hi[k]), lo, sz, rl, n,
                                        self._monotone(lev))
        try:
            graph = dtab.build(int(splits[team.rank]), int(splits[team.rank + 1]), shared=True)
        finally:
            dtab.free()
        try:
            if gs is not None:
                graph = graph.contract(gs)
            maps, _ = scs._fiedler_checked(graph, None, scs.DEFAULT_TOL, scs.DEFAULT_MAX_ITER, 0)
        finally:
            graph.free()
        return maps

    # ------------------------------------------------------------------ pieces of a level
    def _node_arrays(self, lev: Level, k: int) -> TreeArrays:
        """Node ``k`` of ``lev`` as host arrays (taxa numbered 0 .. u_sz - 1): the node-by-node path's input."""
        node_off, parent, taxon, length, support, weights = lev.forest.download(int(lev.t_lo[k]), int(lev.t_hi[k]))
        lo, sz = int(lev.u_lo[k]), int(lev.u_sz[k])
        taxon = np.where(taxon >= 0, taxon - lo, -1).astype(np.int32)
        gids = lev.gid[lo:lo + sz]
        sub = self.sub
        out = TreeArrays(n_taxa=sz, node_off=node_off, parent=parent, taxon=taxon, length=length, support=support,
                         weights=weights, taxa=sub.taxa, ids=gids if sub.ids is None else np.asarray(sub.ids)[gids])
        out.resident_device = sub.resident_device
        return out

    def _graft(self, lev: Level, k: int) -> TreeNode:
        return self._node_arrays(lev, k).to_tree(0)  # reference: scs.py:96-98

    def _exact_groups(self, lev, k, prank, vertex_of, relabel, n_groups, gs_patch) -> None:
        """Two taxa of node ``k`` carry the same signature: its contraction groups from the exact host
        routine on the node's tables (reference: scs.py:302-316; ``flatten.contraction_groups``)."""
        stats["exact_group_nodes"] += 1
        lo, sz = int(lev.u_lo[k]), int(lev.u_sz[k])
        tree_off, leaf_taxon, adj_depth, adj_val, tree_w = lev.forest.tables_range(int(lev.t_lo[k]), int(lev.t_hi[k]))
        n = int(lev.n_pres[k])
        local = prank[leaf_taxon].astype(np.int32)  # (every leaf's taxon is present)
        tables = fl.TreeTables(n_taxa=n, tree_off=tree_off, leaf_taxon=local, adj_depth=adj_depth, adj_val=adj_val,
                               tree_w=tree_w)
        groups = fl.contraction_groups(tables)
        ng = int(groups.max()) + 1
        if ng == n:
            return  # (the signatures collided, the sets do not)
        order = np.lexsort((np.arange(n), groups))  # by group, then id: scs.relabel_for_contraction
        new_of_old = np.empty(n, dtype=np.int32)
        new_of_old[order] = np.arange(n, dtype=np.int32)
        counts = np.bincount(groups, minlength=ng)
        group_start = np.zeros(ng + 1, dtype=np.int32)
        np.cumsum(counts, out=group_start[1:])
        seg = slice(lo, lo + sz)
        here = lev.present[seg].astype(bool)
        rl = relabel[seg]
        rl[here] = new_of_old[prank[seg][here]]
        vo = vertex_of[seg]
        vo[here] = groups[prank[seg][here]]
        n_groups[k] = ng
        gs_patch[k] = group_start
        # members of every vertex as ids among the node's present taxa (perm[group_start[g] : group_start[g + 1]])
        lev.members[k] = (order.astype(np.int32), group_start)

    def _group_starts(self, nodes, n_groups, gs_patch) -> np.ndarray:
        ng1 = n_groups[nodes].astype(np.int64) + 1
        starts = np.cumsum(ng1) - ng1
        out = (np.arange(int(ng1.sum())) - np.repeat(starts, ng1)).astype(np.int32)
        for i, k in enumerate(nodes):
            patch = gs_patch.get(int(k))
            if patch is not None:
                out[int(starts[i]):int(starts[i]) + len(patch)] = patch
        return out

    def _solve_small(self, lev, small, m, relabel, nid, gs_patch) -> None:
        """``scs_small_solve_begin_level`` over the level's small spectral nodes, in batches bounded by the
        addend scratch (trees x cells per node)."""
        cap = int(float(_env.probe("SCS_SPEC_BATCH_GB", "8")) * (1 << 30))
        cost = lev.n_pres[small].astype(np.int64) ** 2 * ((m[small] + 1) & ~1) * 8
        at = 0
        while at < len(small):
            end, used = at, 0
            while end < len(small) and (end == at or used + cost[end] <= cap):
                used += int(cost[end])
                end += 1
            nodes = small[at:end]
            at = end
            sel = np.zeros(lev.K, dtype=bool)
            sel[nodes] = True
            ticket = self.dev.small_solve_begin_level(
                lev.forest, lev.t_lo[nodes], (lev.t_hi - lev.t_lo)[nodes], lev.n_leaves[nodes], lev.u_lo[nodes],
                lev.u_sz[nodes], relabel[sel[nid]], lev.n_pres[nodes], lev.n_groups[nodes],
                self._group_starts(nodes, lev.n_groups, gs_patch))
            maps, lam = ticket.raw()
            ok = np.all(np.isfinite(lam), axis=1)
            ng = lev.n_groups[nodes].astype(np.int64)
            src = np.cumsum(ng) - ng
            if ok.all() and len(nodes) and np.all(np.diff(nodes) == 1):
                lev.maps[int(lev.v_off[nodes[0]]):int(lev.v_off[nodes[-1] + 1])] = maps
            else:
                for i, k in enumerate(nodes):
                    if ok[i]:
                        lev.maps[int(lev.v_off[k]):int(lev.v_off[k + 1])] = maps[int(src[i]):int(src[i] + ng[i])]
            for k in nodes[~ok]:
                lev.kind[k] = FALLBACK  # the one-sided Jacobi ran out of sweeps: this node goes node by node

    def _large_job(self, lev, k, relabel, gs_patch):
        """A node above the batched path's size: build, contract, LOBPCG on its slice of the level's tables --
        as a job for any context on this GPU (``ahead.Ahead``); returns the embedding."""
        from spectralclustersupertree_amd import scs

        lo, sz = int(lev.u_lo[k]), int(lev.u_sz[k])
        forest, t_lo, t_hi = lev.forest, int(lev.t_lo[k]), int(lev.t_hi[k])
        rl = relabel[lo:lo + sz].copy()
        n, mono, gs = int(lev.n_pres[k]), self._monotone(lev), gs_patch.get(k)

        def job(dev):
            t_job = time.perf_counter()
            try:
                return run(dev)
            finally:
                if n >= 4096:
                    stats.setdefault("big_jobs", []).append((n, round(time.perf_counter() - t_job, 3)))

        def run(dev):
            dtab = dev.upload_range(forest, t_lo, t_hi, lo, sz, rl, n, mono)
            try:
                graph = dtab.build()
            finally:
                dtab.free()
            try:
                if gs is not None:
                    graph = graph.contract(gs)
                maps, _ = scs._fiedler_checked(graph, None, scs.DEFAULT_TOL, scs.DEFAULT_MAX_ITER, 0)
            finally:
                graph.free()
            return maps

        return job

    # ------------------------------------------------------------------ the walk proper
    def _names(self, lev: Level, uids) -> list[str]:
        name = self.sub.name
        gid = lev.gid
        return [name(int(gid[int(u)])) for u in uids]

    def _sequential(self, lev: Level, k: int, random_state, parts=None) -> TreeNode:
        """Node ``k`` on the node-by-node path: from its top (``parts`` None) or from its children down."""
        from spectralclustersupertree_amd import scs

        stats["fallbacks"] += 1
        arrays = self._node_arrays(lev, k)
        if parts is None:
            return scs._construct_node(arrays, self.strategy, self.contract, random_state, None, self.team, None,
                                       self.ahead)
        return scs._construct_children(arrays, parts, self.strategy, self.contract, random_state, None, self.team,
                                       self.ahead)

    def redo(self, lev: Level, k: int, labels: np.ndarray) -> "Engine":
        """Spectral node ``k`` of ``lev`` once more from ITS OWN trees with the labels of record: a new engine
        whose first level is that node alone (a slice of the level's forest, nothing copied) with ``labels`` in
        the place of the provisional ones, and everything below it level by level again."""
        e = Engine(self.sub, self.strategy, self.contract, self.dev, self.team, self.ahead)
        e.prov_rs = self.prov_rs
        lo, sz = int(lev.u_lo[k]), int(lev.u_sz[k])
        v0, v1 = int(lev.v_off[k]), int(lev.v_off[k + 1])
        l0 = Level()
        l0.forest = lev.forest.slice(int(lev.t_lo[k]), int(lev.t_hi[k]))
        l0.shift = lo + lev.shift if lev.forest.n_taxa != lev.T else lo
        l0.monotone = lev.monotone
        l0.K, l0.T = 1, sz
        l0.t_lo = np.zeros(1, dtype=np.int32)
        l0.t_hi = np.asarray([int(lev.t_hi[k] - lev.t_lo[k])], dtype=np.int32)
        l0.n_leaves = lev.n_leaves[k:k + 1].copy()
        l0.u_lo = np.zeros(1, dtype=np.int32)
        l0.u_sz = np.asarray([sz], dtype=np.int32)
        l0.gid = lev.gid[lo:lo + sz]
        l0.present = lev.present[lo:lo + sz]
        l0.comp_root = l0.sig = None
        l0.kind = np.asarray([SPECTRAL], dtype=np.int8)
        l0.n_pres = lev.n_pres[k:k + 1].copy()
        l0.n_groups = lev.n_groups[k:k + 1].copy()
        l0.v_off = np.asarray([0, v1 - v0], dtype=np.int64)
        l0.maps = lev.maps[v0:v1]
        l0.prov = np.asarray(labels, dtype=np.int8)
        l0.graft = {}
        l0.defer = np.zeros(1, dtype=bool)
        mem = lev.members.get(k)
        l0.members = {} if mem is None else {0: mem}
        # vertex of every present taxon
        here = np.flatnonzero(l0.present)
        vertex_of = np.arange(len(here), dtype=np.int64)
        if mem is not None:
            perm, group_start = mem
            vertex_of[perm] = np.repeat(np.arange(len(group_start) - 1), np.diff(group_start))
        tpart = np.full(sz, -1, dtype=np.int64)
        tpart[here] = l0.prov[vertex_of]
        stats["levels"] += 1
        e.levels.append(l0)
        nxt = e._split(l0, tpart, np.zeros(sz, dtype=np.int32))
        while nxt is not None:
            e.levels.append(nxt)
            stats["levels"] += 1
            stats["nodes"] += nxt.K
            nxt = e._process(nxt)
        return e

    def build(self, li: int, k: int, random_state) -> TreeNode:
        """The subtree of node ``k`` of level ``li`` in the reference's order of visits and draws
        (reference: scs.py:96-174)."""
        from spectralclustersupertree_amd import kmeans2, scs

        lev = self.levels[li]
        kind = int(lev.kind[k])
        lo, sz = int(lev.u_lo[k]), int(lev.u_sz[k])
        if kind == GRAFT:
            return lev.graft[k]
        if kind == TIPS:
            here = lo + np.flatnonzero(lev.present[lo:lo + sz])
            return tip_names_to_tree(self._names(lev, here))
        if kind == FALLBACK:
            return self._sequential(lev, k, random_state)
        s0, s1 = int(lev.node_seg[k]), int(lev.node_seg[k + 1])
        segs = list(range(s0, s1))
        if kind == SPECTRAL:
            v0, v1 = int(lev.v_off[k]), int(lev.v_off[k + 1])
            lazy = getattr(lev, "lazy", None)
            was_lazy = lazy is not None and lazy.pending(k)
            if was_lazy:
                t0 = time.perf_counter()
                lev.maps[v0:v1] = lazy.fetch(k, self.dev)
                stats["t_lazy_wait"] += time.perf_counter() - t0
            maps = lev.maps[v0:v1]
            # the reference's draws: the ARPACK start vector (sklearn/utils/_arpack.py:31-33), then k-means
            random_state.uniform(-1, 1, v1 - v0)
            labels = np.asarray(kmeans2.labels(maps, random_state))
            prov = lev.prov[v0:v1]
            if scs._node_trace is not None:
                self._trace(lev, k, labels, maps)
            deferred = bool(lev.defer[k])
            if deferred:
                # (nothing was computed below this node: the labels of record decide now)
                # (a node embedded behind the level -- ``_Lazy`` -- was given no provisional labels to agree with)
                if not was_lazy and (np.array_equal(labels, prov) or np.array_equal(labels, 1 - prov)):
                    stats["deferred_agree"] += 1
            elif np.array_equal(labels, prov):
                pass
            elif np.array_equal(labels, 1 - prov):
                segs.reverse()  # (both labels occur, or the arrays would be equal: two segments)
                if len(segs) == 1:  # (all vertices in one cluster: nothing to swap)
                    segs = list(range(s0, s1))
            else:
                deferred = True
                # the draws decided a tie differently: the subtree below this node once more, from the node's own
                # trees with the labels of record
                stats["mismatches"] += 1
                stats["mismatch_sizes"].append(int(v1 - v0))
            if deferred:
                t0 = time.perf_counter()
                try:
                    again = self.redo(lev, k, labels)
                except nv.ScsError as exc:
                    if exc.code != nv.ENOMEM or self._spread():
                        raise
                    again = None
                stats["t_redo"] += time.perf_counter() - t0
                stats["redo_log"].append((int(v1 - v0), time.perf_counter() - t0, 0 if again is None else len(again.levels)))
                if again is None:  # (no room on the device: node by node, from host arrays)
                    present_local = np.flatnonzero(lev.present[lo:lo + sz])
                    parts: list[list[int]] = [[], []]
                    for g, lab in enumerate(labels):
                        parts[int(lab)].extend(int(present_local[int(i)]) for i in self._members(lev, k, g))
                    return self._sequential(lev, k, random_state, parts)
                first = again.levels[0]
                return again._children(0, list(range(int(first.node_seg[0]), int(first.node_seg[1]))), random_state)
        return self._children(li, segs, random_state)

    def _children(self, li: int, segs, random_state) -> TreeNode:
        """The subtrees of a node's parts (segments ``segs`` of level ``li``, in the order of the visit) joined
        under a new root (reference: scs.py:139-174)."""
        lev = self.levels[li]
        nxt = self.levels[li + 1] if li + 1 < len(self.levels) else None
        for s in segs:  # reference: scs.py:63-65 reached from :158 -- before any child is entered
            c = int(lev.seg_child[s])
            if c >= 0 and int(nxt.kind[c]) == EMPTY:
                msg = "There must be at least one tree to make a supertree."
                raise ValueError(msg)
        child_trees: list = []
        for s in segs:
            a = int(lev.seg_start[s])
            uids = lev.sorted_taxa[a:a + int(lev.seg_len[s])]
            c = int(lev.seg_child[s])
            if c < 0:
                if len(uids) > 2:  # a part of a node that left the engine (more than eight parts)
                    msg = "levels: a part without a child"
                    raise AssertionError(msg)
                child_trees.append(tip_names_to_tree(self._names(lev, uids)))
                continue
            child_trees.append(self.build(li + 1, c, random_state))
            clo, csz = int(nxt.u_lo[c]), int(nxt.u_sz[c])
            missing = clo + np.flatnonzero(nxt.present[clo:clo + csz] == 0)
            if len(missing):  # taxa no surviving tree holds (scs.py:166-170)
                child_trees.extend(TreeNode(x) for x in self._names(nxt, missing))
        return connect_trees(child_trees)

    def _members(self, lev: Level, k: int, g: int):
        """Vertex ``g`` of spectral node ``k``: ids among the node's present taxa."""
        mem = lev.members.get(k)
        if mem is None:
            return (g,)
        perm, group_start = mem
        return perm[int(group_start[g]):int(group_start[g + 1])]

    def _trace(self, lev: Level, k: int, labels, maps) -> None:
        from spectralclustersupertree_amd import scs

        lo, sz = int(lev.u_lo[k]), int(lev.u_sz[k])
        here = lo + np.flatnonzero(lev.present[lo:lo + sz])
        names = self._names(lev, here)
        vertices = [tuple(names[int(i)] for i in self._members(lev, k, g)) for g in range(len(labels))]
        scs._node_trace.append({"vertices": vertices, "labels": labels.copy(), "maps": np.array(maps, copy=True)})
