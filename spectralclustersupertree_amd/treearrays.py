"""Source trees as flat node arrays: the recursion of ``construct_supertree`` without
Python tree objects below the top level.

The reference restricts every source tree to a taxon subset at every node of the
recursion (``_generate_induced_trees_with_weights``, reference:
src/sc_supertree/scs.py:411-455) by walking cogent3 tree objects.  Here the trees are
converted ONCE into preorder node arrays; restriction (drop, splice, collapse the
root, drop trees with fewer than two leaves) and the flattening into the device
tables of ``include/scs_hip.h`` run in ``libscs_host.so`` (plain C,
``csrc/scs_host.c``) in time linear in the number of nodes, producing bit for bit the
tables the object path (``tree.TreeNode.get_sub_tree`` + ``flatten.flatten_trees``)
produces -- ``tests/test_treearrays.py`` checks exactly that.
"""

from __future__ import annotations

import ctypes as C
from collections.abc import Sequence
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from spectralclustersupertree_amd._hostlib import load as _load
from spectralclustersupertree_amd.flatten import STRATEGIES, TreeTables
from spectralclustersupertree_amd.tree import TreeNode

_STRATEGY_CODE = {"one": 0, "depth": 1, "branch": 2, "bootstrap": 3}  # csrc/scs_host.c
_ERRORS = {-1: "out of memory", -2: "malformed tree arrays", -3: "missing support"}


_NP_OF = {C.c_int32: np.int32, C.c_int64: np.int64, C.c_double: np.float64, C.c_uint8: np.uint8}


def _p(a: np.ndarray, ct):
    """Address of a contiguous array of the C type `ct` (the bindings take plain addresses)."""
    assert a.dtype == _NP_OF[ct] and a.flags.c_contiguous
    return a.ctypes.data


@dataclass
class TreeArrays:
    """A forest over integer taxon ids ``0..n_taxa-1`` in preorder node arrays."""

    n_taxa: int
    node_off: np.ndarray  # int64 [M+1]
    parent: np.ndarray  # int32, relative to the tree's first node, -1 = root
    taxon: np.ndarray  # int32, -1 = internal node
    length: np.ndarray  # float64, NaN = None
    support: np.ndarray  # float64, NaN = None
    weights: np.ndarray  # float64 [M]
    taxa: list[str]  # name table (ids are ranks of the names in sorted order)
    # taxon id of THIS forest -> index into ``taxa``; None = identity.  A child produced by
    # ``split`` numbers its taxa 0..k-1 (in the order of the parent's ids, i.e. by name), so
    # nothing a deep recursion node does is proportional to the size of the whole input.
    ids: np.ndarray | None = None
    _present: np.ndarray | None = None  # cached result of present_taxa (filled by split)
    _leaf_counts: np.ndarray | None = None
    # the device the recursion may move this forest to (backend.Device; None: splits stay on the host)
    resident_device: object | None = None

    @property
    def n_trees(self) -> int:
        return len(self.weights)

    def name(self, i: int) -> str:
        """Name of taxon ``i`` of this forest."""
        return self.taxa[int(i) if self.ids is None else int(self.ids[int(i)])]

    # ------------------------------------------------------------------ build
    @classmethod
    def from_trees(cls, trees: Sequence, weights: Sequence[float], taxa: Sequence[str]) -> "TreeArrays":
        """One preorder walk per tree object (duck-typed: iteration over children,
        ``is_tip``, ``name``, ``length``, ``support``)."""
        index = {name: i for i, name in enumerate(taxa)}
        node_off = [0]
        parent: list[int] = []
        taxon: list[int] = []
        length: list[float] = []
        support: list[float] = []
        nan = float("nan")
        for tree in trees:
            base = len(parent)
            stack = [(tree, -1)]
            while stack:
                node, par = stack.pop()
                me = len(parent) - base
                parent.append(par)
                tip = node.is_tip()
                taxon.append(index[node.name] if tip else -1)
                ln = getattr(node, "length", None)
                sp = getattr(node, "support", None)
                length.append(nan if ln is None else float(ln))
                support.append(nan if sp is None else float(sp))
                if not tip:
                    kids = list(node)
                    for child in reversed(kids):
                        stack.append((child, me))
            node_off.append(len(parent))
        return cls(
            n_taxa=len(taxa),
            node_off=np.asarray(node_off, dtype=np.int64),
            parent=np.asarray(parent, dtype=np.int32),
            taxon=np.asarray(taxon, dtype=np.int32),
            length=np.asarray(length, dtype=np.float64),
            support=np.asarray(support, dtype=np.float64),
            weights=np.asarray([float(w) for w in weights], dtype=np.float64),
            taxa=list(taxa),
        )

    @classmethod
    def from_newick_file(cls, path, weights: Sequence[float] | None = None) -> "TreeArrays":
        """Parse a line-separated Newick file straight into arrays (reference: load.py:7-23;
        the grammar and its meaning are those of ``tree.make_tree``), in C, without building
        tree objects.  Taxon ids are the ranks of the sorted distinct leaf names."""
        lib = _load()
        text = Path(path).read_bytes()
        n_trees, n_nodes, name_bytes, max_nodes, err = (C.c_int64(), C.c_int64(), C.c_int64(),
                                                        C.c_int64(), C.c_int64())
        rc = lib.scs_host_newick_scan(text, len(text), C.byref(n_trees), C.byref(n_nodes),
                                      C.byref(name_bytes), C.byref(max_nodes), C.byref(err))
        if rc:
            raise ValueError(f"malformed Newick tree on line {err.value + 1} of {path}")
        m, k = n_trees.value, n_nodes.value
        node_off = np.zeros(m + 1, dtype=np.int64)
        parent = np.empty(k, dtype=np.int32)
        length = np.empty(k, dtype=np.float64)
        support = np.empty(k, dtype=np.float64)
        name_off = np.empty(k, dtype=np.int64)
        pool = C.create_string_buffer(name_bytes.value + 1)
        used = C.c_int64()
        rc = lib.scs_host_newick_parse(text, len(text), m, max_nodes.value, _p(node_off, C.c_int64),
                                       _p(parent, C.c_int32), _p(length, C.c_double),
                                       _p(support, C.c_double), _p(name_off, C.c_int64), pool,
                                       C.byref(used), C.byref(err))
        if rc:
            raise ValueError(f"malformed Newick tree on line {err.value + 1} of {path}")
        taxon = np.empty(k, dtype=np.int32)
        uniq = np.empty(max(k, 1), dtype=np.int64)
        n_taxa = C.c_int64()
        rc = lib.scs_host_names_rank(pool, _p(name_off, C.c_int64), k, _p(taxon, C.c_int32),
                                     _p(uniq, C.c_int64), C.byref(n_taxa))
        if rc:
            raise MemoryError("scs_host_names_rank")
        raw = pool.raw
        taxa = []
        for off in uniq[: n_taxa.value]:
            end = raw.index(b"\0", int(off))
            taxa.append(raw[int(off):end].decode())
        if weights is None:
            weights = np.ones(m, dtype=np.float64)
        return cls(n_taxa=n_taxa.value, node_off=node_off, parent=parent, taxon=taxon, length=length,
                   support=support, weights=np.asarray(weights, dtype=np.float64), taxa=taxa)

    # ---------------------------------------------------------------- queries
    def present_taxa(self) -> np.ndarray:
        """Sorted ids of the taxa that occur in at least one tree."""
        if self._present is not None:
            return self._present
        mark = np.zeros(max(self.n_taxa, 1), dtype=np.uint8)
        if self.n_trees:
            _load().scs_host_present(self.n_trees, _p(self.node_off, C.c_int64), _p(self.taxon, C.c_int32),
                                     _p(mark, C.c_uint8))
        self._present = np.flatnonzero(mark[: self.n_taxa]).astype(np.int32)
        return self._present

    def leaf_counts(self) -> np.ndarray:
        if self._leaf_counts is not None:
            return self._leaf_counts
        out = np.zeros(self.n_trees, dtype=np.int64)
        if self.n_trees:
            _load().scs_host_leaf_counts(self.n_trees, _p(self.node_off, C.c_int64),
                                         _p(self.taxon, C.c_int32), _p(out, C.c_int64))
        self._leaf_counts = out
        return out

    # ------------------------------------------------------------ restriction
    def restrict(self, keep_ids: np.ndarray) -> "TreeArrays":
        """Forest induced on the taxa ``keep_ids`` (reference: scs.py:411-455)."""
        lib = _load()
        keep = np.zeros(self.n_taxa, dtype=np.uint8)
        keep[np.asarray(keep_ids, dtype=np.int64)] = 1
        m = self.n_trees
        tree_keep = np.zeros(max(m, 1), dtype=np.uint8)
        nodes = np.zeros(max(m, 1), dtype=np.int32)
        if m:
            rc = lib.scs_host_restrict_sizes(m, _p(self.node_off, C.c_int64), _p(self.parent, C.c_int32),
                                             _p(self.taxon, C.c_int32), _p(keep, C.c_uint8),
                                             _p(tree_keep, C.c_uint8), _p(nodes, C.c_int32))
            if rc:
                raise ValueError(f"scs_host_restrict_sizes: {_ERRORS.get(rc, rc)}")
        kept = np.flatnonzero(tree_keep[:m])
        new_off = np.zeros(len(kept) + 1, dtype=np.int64)
        np.cumsum(nodes[:m][kept], out=new_off[1:])
        total = int(new_off[-1])
        out = TreeArrays(
            n_taxa=self.n_taxa,
            node_off=new_off,
            parent=np.empty(total, dtype=np.int32),
            taxon=np.empty(total, dtype=np.int32),
            length=np.empty(total, dtype=np.float64),
            support=np.empty(total, dtype=np.float64),
            weights=self.weights[kept].copy(),
            taxa=self.taxa,
            ids=self.ids,
        )
        if len(kept):
            rc = lib.scs_host_restrict_fill(
                m, _p(self.node_off, C.c_int64), _p(self.parent, C.c_int32), _p(self.taxon, C.c_int32),
                _p(self.length, C.c_double), _p(self.support, C.c_double), _p(keep, C.c_uint8),
                _p(tree_keep, C.c_uint8), _p(new_off, C.c_int64), _p(out.parent, C.c_int32),
                _p(out.taxon, C.c_int32), _p(out.length, C.c_double), _p(out.support, C.c_double))
            if rc:
                raise ValueError(f"scs_host_restrict_fill: {_ERRORS.get(rc, rc)}")
        return out

    def split(self, parts: Sequence[np.ndarray], strategy: str | None = None) -> list["TreeArrays"]:
        """The forests induced on each of the disjoint taxon sets ``parts`` (sorted id arrays),
        all from ONE sweep of this forest (``scs_host_split_*``; reference: the loop over the
        parts at scs.py:139-155 with the restriction of :411-455).  Child ``c`` numbers its taxa
        ``0..len(parts[c])-1`` in the order of ``parts[c]``; its ``present_taxa`` and
        ``leaf_counts`` come for free.

        With a device attached (``resident_device``, set by the recursion) and a ``strategy`` the
        forest goes to HBM once and the split -- and every later one below it -- runs there
        (``ResidentArrays``, ``scs_forest_split``): the same child forests and tables, bit for bit."""
        lib = _load()
        n_parts = len(parts)
        if n_parts == 0:
            return []
        if (self.resident_device is not None and strategy is not None and resident_split_wanted(self, n_parts)):
            from spectralclustersupertree_amd import _native as nv

            dev = self.resident_device() if callable(self.resident_device) else self.resident_device
            try:
                return ResidentArrays.from_host(self, dev).split(parts, strategy)
            except nv.ScsError as exc:
                if exc.code != nv.ENOMEM:
                    raise
                # not enough device memory for the forest and its children beside what the solves hold:
                # this split runs on the host (the children carry the device on and may move later)
        part_of = np.full(max(self.n_taxa, 1), -1, dtype=np.int32)
        new_id = np.zeros(max(self.n_taxa, 1), dtype=np.int32)
        parts = [np.asarray(p, dtype=np.int32) for p in parts]
        for c, ids in enumerate(parts):
            part_of[ids] = c
            new_id[ids] = np.arange(len(ids), dtype=np.int32)
        plan = C.c_void_p()
        part_trees = np.zeros(n_parts, dtype=np.int64)
        part_nodes = np.zeros(n_parts, dtype=np.int64)
        leaf_counts = np.ascontiguousarray(self.leaf_counts(), dtype=np.int64)
        if len(leaf_counts) == 0:
            leaf_counts = np.zeros(1, dtype=np.int64)
        rc = lib.scs_host_split_begin(self.n_trees, _p(self.node_off, C.c_int64), _p(self.parent, C.c_int32),
                                      _p(self.taxon, C.c_int32), _p(self.length, C.c_double),
                                      _p(self.support, C.c_double), _p(leaf_counts, C.c_int64),
                                      _p(part_of, C.c_int32),
                                      _p(new_id, C.c_int32), n_parts, C.byref(plan),
                                      _p(part_trees, C.c_int64), _p(part_nodes, C.c_int64))
        if rc:
            raise ValueError(f"scs_host_split_begin: {_ERRORS.get(rc, rc)}")
        out = []
        try:
            for c, ids in enumerate(parts):
                m, total = int(part_trees[c]), int(part_nodes[c])
                node_off = np.zeros(m + 1, dtype=np.int64)
                tree_index = np.empty(max(m, 1), dtype=np.int32)
                leaf_counts = np.zeros(max(m, 1), dtype=np.int64)
                present = np.zeros(max(len(ids), 1), dtype=np.uint8)
                child = TreeArrays(
                    n_taxa=len(ids), node_off=node_off,
                    parent=np.empty(total, dtype=np.int32), taxon=np.empty(total, dtype=np.int32),
                    length=np.empty(total, dtype=np.float64), support=np.empty(total, dtype=np.float64),
                    weights=np.empty(0, dtype=np.float64), taxa=self.taxa,
                    ids=ids if self.ids is None else self.ids[ids])
                if m:
                    rc = lib.scs_host_split_fill(plan, c, _p(node_off, C.c_int64), _p(tree_index, C.c_int32),
                                                 _p(leaf_counts, C.c_int64), _p(child.parent, C.c_int32),
                                                 _p(child.taxon, C.c_int32), _p(child.length, C.c_double),
                                                 _p(child.support, C.c_double), _p(present, C.c_uint8))
                    if rc:
                        raise ValueError(f"scs_host_split_fill: {_ERRORS.get(rc, rc)}")
                    child.weights = self.weights[tree_index[:m]].copy()
                child._present = np.flatnonzero(present[: len(ids)]).astype(np.int32)
                child._leaf_counts = leaf_counts[:m]
                child.resident_device = self.resident_device  # (a big child may move to the device later)
                out.append(child)
        finally:
            lib.scs_host_split_end(plan)
        return out

    # -------------------------------------------------------------- flattening
    def flatten(self, strategy: str, local_ids: np.ndarray | None = None) -> TreeTables:
        """Device tables of the forest (what ``flatten.flatten_trees`` builds from objects).

        ``local_ids`` (sorted global ids) renumbers the taxa to ``0..len-1`` -- a
        recursion node numbers its own taxa by sorted name, as the reference does.
        """
        if strategy not in STRATEGIES:
            msg = f"Invalid weighting strategy selected: '{strategy}'"
            raise ValueError(msg)
        lib = _load()
        m = self.n_trees
        leaf_off = np.zeros(m + 1, dtype=np.int64)
        np.cumsum(self.leaf_counts(), out=leaf_off[1:])
        total = int(leaf_off[-1])
        leaf_taxon = np.empty(total, dtype=np.int32)
        adj_depth = np.empty(total, dtype=np.int32)
        adj_val = np.empty(total, dtype=np.float64)
        mono = C.c_int32(1)
        lut = None
        if local_ids is not None:
            local_ids = np.asarray(local_ids, dtype=np.int32)
            lut = np.zeros(max(self.n_taxa, 1), dtype=np.int32)  # global id -> position in local_ids
            lut[local_ids] = np.arange(len(local_ids), dtype=np.int32)
        if m:
            rc = lib.scs_host_flatten(m, _p(self.node_off, C.c_int64), _p(self.parent, C.c_int32),
                                      _p(self.taxon, C.c_int32), _p(self.length, C.c_double),
                                      _p(self.support, C.c_double), _STRATEGY_CODE[strategy],
                                      _p(leaf_off, C.c_int64), _p(leaf_taxon, C.c_int32),
                                      _p(adj_depth, C.c_int32), _p(adj_val, C.c_double), C.byref(mono),
                                      _p(lut, C.c_int32) if lut is not None else None)
            if rc == -3:
                # the reference fails in ``length * tree_weight`` with a missing support
                # (reference: scs.py:656)
                msg = "unsupported operand type(s) for *: 'NoneType' and 'float'"
                raise TypeError(msg)
            if rc:
                raise ValueError(f"scs_host_flatten: {_ERRORS.get(rc, rc)}")
        n_taxa = self.n_taxa
        taxa = self.taxa if self.ids is None else [self.name(i) for i in range(n_taxa)]
        if local_ids is not None:
            n_taxa = len(local_ids)
            taxa = [self.name(i) for i in local_ids]
        monotone = (strategy in ("one", "depth", "branch") and bool(mono.value)
                    and bool(np.all(self.weights >= 0)))
        return TreeTables(n_taxa=n_taxa, tree_off=leaf_off, leaf_taxon=leaf_taxon, adj_depth=adj_depth,
                          adj_val=adj_val, tree_w=self.weights.copy(), taxa=taxa, monotone=monotone)

    # --------------------------------------------------------------- to object
    def to_tree(self, t: int) -> TreeNode:
        """Tree ``t`` as a ``TreeNode`` (internal names are not kept)."""
        lo, hi = int(self.node_off[t]), int(self.node_off[t + 1])
        nodes: list[TreeNode] = []
        for i in range(lo, hi):
            tx = int(self.taxon[i])
            ln, sp = float(self.length[i]), float(self.support[i])
            node = TreeNode(self.name(tx) if tx >= 0 else "", None,
                            None if ln != ln else ln, None if sp != sp else sp)
            nodes.append(node)
            par = int(self.parent[i])
            if par >= 0:
                nodes[par].children.append(node)
                node.parent = nodes[par]
        return nodes[0]


# ---------------------------------------------------------------------------
# forests resident in HBM (round 5): the recursion's restriction step on the device
# ---------------------------------------------------------------------------
def resident_split_wanted(forest, n_parts: int) -> bool:
    """Whether a split of ``forest`` into ``n_parts`` runs on the device: at most 8 parts (one
    mark bit per part), and enough tree nodes that three launches and two round trips are
    cheaper than the host's sweep (SCS_DEVICE_SPLIT=0 switches the path off,
    SCS_DEVICE_SPLIT_MIN_NODES moves the threshold)."""
    from spectralclustersupertree_amd import _env

    if not int(_env.probe("SCS_DEVICE_SPLIT", "1")):
        return False
    if n_parts > 8 or forest.n_trees == 0:
        return False
    n_nodes = forest.n_nodes if isinstance(forest, ResidentArrays) else len(forest.parent)
    if n_nodes < int(_env.probe("SCS_DEVICE_SPLIT_MIN_NODES", "20000")):
        return False
    # (small trees: one thread per tree on an LDS copy; big ones: every step per node -- scs_forest.hip;
    # SCS_DEVICE_SPLIT_MAX_TREE_NODES keeps forests of bigger trees on the host: measurements)
    cap = int(_env.probe("SCS_DEVICE_SPLIT_MAX_TREE_NODES", "0"))
    return cap <= 0 or n_nodes <= forest.n_trees * cap


class ResidentArrays:
    """A forest whose node arrays live in HBM (``backend.DeviceForest``), with the interface the
    recursion uses of ``TreeArrays``: ``n_trees``, ``name``, ``present_taxa``, ``flatten``,
    ``split``, ``to_tree``.  A child of ``split`` arrives with its tables already flattened on
    the device (for the weighting strategy the split was asked for) and downloaded; the node
    arrays themselves never travel unless ``to_host`` is called (a split into more than eight
    parts, a forest too small to be worth a launch, a single tree to graft)."""

    def __init__(self, forest, n_taxa: int, weights: np.ndarray, taxa: list[str], ids: np.ndarray | None,
                 strategy: str | None = None, tables=None) -> None:
        self.forest = forest  # backend.DeviceForest
        self.n_taxa = int(n_taxa)
        self.weights = weights
        self.taxa = taxa
        self.ids = ids
        self.strategy = strategy
        self._tables = tables  # (tree_off, leaf_taxon, adj_depth, adj_val) of a split's child
        self._present: np.ndarray | None = None
        self.resident_device = forest.dev

    # ---- TreeArrays' interface ------------------------------------------------
    @property
    def n_trees(self) -> int:
        return self.forest.n_trees

    @property
    def n_nodes(self) -> int:
        return self.forest.n_nodes

    def name(self, i: int) -> str:
        return self.taxa[int(i) if self.ids is None else int(self.ids[int(i)])]

    @classmethod
    def from_host(cls, arrays: "TreeArrays", dev) -> "ResidentArrays":
        from spectralclustersupertree_amd.backend import DeviceForest

        forest = DeviceForest.upload(
            dev, max(arrays.n_taxa, 1), np.ascontiguousarray(arrays.node_off, dtype=np.int64),
            np.ascontiguousarray(arrays.parent, dtype=np.int32), np.ascontiguousarray(arrays.taxon, dtype=np.int32),
            np.ascontiguousarray(arrays.length, dtype=np.float64),
            np.ascontiguousarray(arrays.support, dtype=np.float64),
            np.ascontiguousarray(arrays.weights, dtype=np.float64), int(arrays.leaf_counts().sum()))
        out = cls(forest, arrays.n_taxa, arrays.weights, arrays.taxa, arrays.ids)
        out._present = arrays._present
        return out

    def to_host(self) -> "TreeArrays":
        node_off, parent, taxon, length, support, weights = self.forest.download()
        out = TreeArrays(n_taxa=self.n_taxa, node_off=node_off, parent=parent, taxon=taxon, length=length,
                         support=support, weights=weights, taxa=self.taxa, ids=self.ids)
        out._present = self._present
        out.resident_device = self.resident_device
        return out

    def present_taxa(self) -> np.ndarray:
        if self._present is None:
            self._present = self.to_host().present_taxa()
        return self._present

    def leaf_counts(self) -> np.ndarray:
        if self._tables is not None:
            return np.diff(self._tables[0])
        return self.to_host().leaf_counts()

    def to_tree(self, t: int) -> TreeNode:
        node_off, parent, taxon, length, support, weights = self.forest.download(t, t + 1)
        one = TreeArrays(n_taxa=self.n_taxa, node_off=node_off, parent=parent, taxon=taxon, length=length,
                         support=support, weights=weights, taxa=self.taxa, ids=self.ids)
        return one.to_tree(0)

    def flatten(self, strategy: str, local_ids: np.ndarray | None = None) -> TreeTables:
        """The tables the split flattened on the device (same bits as ``TreeArrays.flatten``)."""
        if strategy not in STRATEGIES:
            msg = f"Invalid weighting strategy selected: '{strategy}'"
            raise ValueError(msg)
        if self._tables is None or strategy != self.strategy:
            return self.to_host().flatten(strategy, local_ids)
        tree_off, leaf_taxon, adj_depth, adj_val = self._tables
        n_taxa = self.n_taxa
        relabel = None
        if local_ids is not None:
            local_ids = np.asarray(local_ids, dtype=np.int32)
            n_taxa = len(local_ids)
            if n_taxa != self.n_taxa or not np.array_equal(local_ids, np.arange(n_taxa, dtype=np.int32)):
                lut = np.zeros(max(self.n_taxa, 1), dtype=np.int32)  # id of this forest -> position in local_ids
                lut[local_ids] = np.arange(n_taxa, dtype=np.int32)
                leaf_taxon = lut[leaf_taxon]
                relabel = lut
            taxa = [self.name(i) for i in local_ids]
        else:
            taxa = self.taxa if self.ids is None else [self.name(i) for i in range(n_taxa)]
        monotone = (strategy in ("one", "depth", "branch") and bool(self.forest.monotone_flag)
                    and bool(np.all(self.weights >= 0)))
        return TreeTables(n_taxa=n_taxa, tree_off=tree_off, leaf_taxon=leaf_taxon, adj_depth=adj_depth,
                          adj_val=adj_val, tree_w=self.weights.copy(), taxa=taxa, monotone=monotone,
                          resident=(self.forest, relabel))

    def split(self, parts: Sequence[np.ndarray], strategy: str | None = None) -> list:
        """``TreeArrays.split`` on the device (``scs_forest_split``; reference: scs.py:139-155 with
        :411-455): children stay resident, their tables come back flattened for ``strategy``."""
        n_parts = len(parts)
        if n_parts == 0:
            return []
        if strategy is None or not resident_split_wanted(self, n_parts):
            return self.to_host().split(parts)  # (host arrays carry the device on: a big grandchild may return)
        if strategy not in STRATEGIES:
            msg = f"Invalid weighting strategy selected: '{strategy}'"
            raise ValueError(msg)
        part_of = np.full(max(self.n_taxa, 1), -1, dtype=np.int32)
        new_id = np.zeros(max(self.n_taxa, 1), dtype=np.int32)
        parts = [np.ascontiguousarray(p, dtype=np.int32) for p in parts]
        for c, ids in enumerate(parts):
            part_of[ids] = c
            new_id[ids] = np.arange(len(ids), dtype=np.int32)
        from spectralclustersupertree_amd import _native as nv

        try:
            kids = self.forest.split(part_of, new_id, [len(ids) for ids in parts], _STRATEGY_CODE[strategy])
        except nv.ScsError as exc:
            if exc.code != nv.ENOMEM:
                raise
            return self.to_host().split(parts)  # (out of device memory: this split on the host)
        out = []
        for c, (ids, forest) in enumerate(zip(parts, kids)):
            tree_off, leaf_taxon, adj_depth, adj_val, tree_index, tree_w, present = forest.tables()
            child = ResidentArrays(forest, len(ids), tree_w, self.taxa, ids if self.ids is None else self.ids[ids],
                                   strategy, (tree_off, leaf_taxon, adj_depth, adj_val))
            child._present = np.flatnonzero(present[: len(ids)]).astype(np.int32)
            out.append(child)
        return out
