"""Flatten rooted source trees into the per-tree tables the device consumes.

For tree ``t`` with ``n_t`` leaves in depth-first order the tables hold

* ``leaf_taxon[p]``  int32  global taxon id of the p-th leaf,
* ``adj_depth[p]``   int32  depth of LCA(leaf p, leaf p+1); the root has depth 0,
* ``adj_val[p]``     fp64   the weighting-strategy value carried by that LCA,
* ``tree_w[t]``      fp64   the tree's weight,

all trees concatenated, tree ``t`` occupying ``[tree_off[t], tree_off[t+1])``
(the last ``adj_*`` slot of every tree is unused padding so one offset array
serves all three).  LCA(leaf a, leaf b), a < b, is the entry of minimum depth
in ``adj_depth[a:b]``; the pair is a *proper cluster* iff that depth is > 0.
This is exactly the information ``_dfs_pcg_weights`` consumes
(reference: src/sc_supertree/scs.py:586-663): the value it multiplies by the
tree weight at an internal node is the ``length`` produced by the strategy's
``length_function`` (reference: src/sc_supertree/scs.py:555-564), started at 0
for every child of the root (reference: src/sc_supertree/scs.py:577).

Also here: the two host-side graph computations that only need the tables --
connected components of the proper cluster graph and the contraction classes
(SURVEY.md section 8a rows A3/A4); tie-breaking and naming stay on the host.
"""

from __future__ import annotations

from collections.abc import Sequence
from dataclasses import dataclass

import numpy as np

STRATEGIES = ("one", "branch", "depth", "bootstrap")


@dataclass
class TreeTables:
    """Flattened source trees over integer taxon ids ``0..n_taxa-1``."""

    n_taxa: int
    tree_off: np.ndarray  # int64 [M+1]
    leaf_taxon: np.ndarray  # int32 [L]
    adj_depth: np.ndarray  # int32 [L]
    adj_val: np.ndarray  # float64 [L]
    tree_w: np.ndarray  # float64 [M]
    taxa: list[str] | None = None  # id -> name (None for synthetic tables)
    # True when the strategy value never decreases from an ancestor to a descendant in any
    # tree (`one`, `depth`, `branch` without negative internal lengths): lets the device take
    # its cheaper monotone kernel (same bits).  False is always safe.
    monotone: bool = False
    # (forest, relabel) when these tables are also RESIDENT on the device: `forest` a
    # backend.DeviceForest (a child of scs_forest_split) whose tables equal these except that its
    # leaf_taxon[p] maps to ours through relabel (int32 array over the forest's taxa, or None =
    # identity).  Lets a small node's solve pack its leaf arrays on the device (Device.small_solve_begin).
    resident: object | None = None

    @property
    def n_trees(self) -> int:
        return len(self.tree_w)

    @property
    def n_leaves(self) -> int:
        return int(self.tree_off[-1])

    def validate(self, ranges: bool = True) -> None:
        """Shape/dtype checks done on the host before any native call; ``ranges`` adds the
        O(leaves) range checks (``Device.upload`` leaves those to the kernel that checks the
        uploaded copy)."""
        m = self.n_trees
        if self.tree_off.dtype != np.int64 or self.tree_off.shape != (m + 1,):
            raise ValueError("tree_off must be int64 of length n_trees + 1")
        if self.tree_off[0] != 0 or np.any(np.diff(self.tree_off) < 1):
            raise ValueError("tree_off must start at 0 and every tree needs >= 1 leaf")
        total = self.n_leaves
        for name, arr, dt in (
            ("leaf_taxon", self.leaf_taxon, np.int32),
            ("adj_depth", self.adj_depth, np.int32),
            ("adj_val", self.adj_val, np.float64),
        ):
            if arr.dtype != dt or arr.shape != (total,) or not arr.flags.c_contiguous:
                raise ValueError(f"{name} must be C-contiguous {np.dtype(dt).name} of length {total}")
        if self.tree_w.dtype != np.float64 or not self.tree_w.flags.c_contiguous:
            raise ValueError("tree_w must be C-contiguous float64")
        if not ranges:
            return
        if total and (self.leaf_taxon.min() < 0 or self.leaf_taxon.max() >= self.n_taxa):
            raise ValueError("leaf_taxon out of range")
        if total and self.adj_depth.min() < 0:
            raise ValueError("adj_depth must be >= 0")


def strategy_value(strategy: str, parent_value, node):
    """The reference's ``length_function`` for one internal node.

    reference: src/sc_supertree/scs.py:555-564
    """
    if strategy == "one":
        return 1
    if strategy == "depth":
        return parent_value + 1
    if strategy == "branch":
        length = node.length
        return parent_value + (1 if length is None else length)
    if strategy == "bootstrap":
        return node.support
    msg = f"Invalid weighting strategy selected: '{strategy}'"
    raise ValueError(msg)


def flatten_trees(
    trees: Sequence,
    weights: Sequence[float],
    strategy: str,
    taxa: Sequence[str] | None = None,
) -> TreeTables:
    """Depth-first flatten of duck-typed tree objects (see ``tree.TreeNode``).

    ``taxa`` fixes the id order; default is the sorted union of tip names (the
    canonical vertex order the spectral step uses).
    """
    if strategy not in STRATEGIES:
        msg = f"Invalid weighting strategy selected: '{strategy}'"
        raise ValueError(msg)
    if taxa is None:
        names: set[str] = set()
        for tree in trees:
            names.update(tree.get_tip_names())
        taxa = sorted(names)
    taxa = list(taxa)
    index = {name: i for i, name in enumerate(taxa)}

    # value * tree_weight must not decrease along any root path: no negative tree weights
    monotone = strategy in ("one", "depth", "branch") and all(float(w) >= 0 for w in weights)
    tree_off = [0]
    leaf_taxon: list[int] = []
    adj_depth: list[int] = []
    adj_val: list[float] = []

    for tree, weight in zip(trees, weights):
        first_leaf = True
        pend_depth, pend_val = 0, 0.0
        # stack entries: [node, depth, value, next child index]
        stack = [[tree, 0, 0, 0]]
        if tree.is_tip():
            leaf_taxon.append(index[tree.name])
            adj_depth.append(0)
            adj_val.append(0.0)
            tree_off.append(len(leaf_taxon))
            continue
        children_cache = {id(tree): list(tree)}
        while stack:
            top = stack[-1]
            node, depth, value, k = top
            kids = children_cache[id(node)]
            if k >= len(kids):
                stack.pop()
                del children_cache[id(node)]
                continue
            top[3] = k + 1
            if k >= 1:
                # the next leaf's LCA with the previous leaf is this node
                pend_depth, pend_val = depth, value
            child = kids[k]
            if child.is_tip():
                if not first_leaf:
                    adj_depth.append(pend_depth)
                    adj_val.append(pend_val)
                first_leaf = False
                leaf_taxon.append(index[child.name])
            else:
                ckids = list(child)
                cval = strategy_value(strategy, value, child)
                if cval is None and len(ckids) >= 2:
                    # the reference fails in ``length * tree_weight`` with a
                    # missing support (reference: scs.py:656)
                    msg = "unsupported operand type(s) for *: 'NoneType' and 'float'"
                    raise TypeError(msg)
                if strategy == "branch" and child.length is not None and child.length < 0:
                    monotone = False
                children_cache[id(child)] = ckids
                stack.append([child, depth + 1, 0 if cval is None else cval, 0])
        # padding slot so adj_* share tree_off with leaf_taxon
        adj_depth.append(0)
        adj_val.append(0.0)
        tree_off.append(len(leaf_taxon))
        if len(adj_depth) != len(leaf_taxon):
            msg = "internal error: table length mismatch"
            raise AssertionError(msg)

    tables = TreeTables(
        n_taxa=len(taxa),
        tree_off=np.asarray(tree_off, dtype=np.int64),
        leaf_taxon=np.asarray(leaf_taxon, dtype=np.int32),
        adj_depth=np.asarray(adj_depth, dtype=np.int32),
        adj_val=np.asarray(adj_val, dtype=np.float64),
        tree_w=np.asarray([float(w) for w in weights], dtype=np.float64),
        taxa=taxa,
        monotone=monotone,
    )
    return tables


# ---------------------------------------------------------------------------
# Host-side graph computations over the tables
# ---------------------------------------------------------------------------


def leaf_side_ids(tables: TreeTables) -> np.ndarray:
    """For every leaf slot, a globally unique id of (tree, root-side child).

    Two leaves of a tree form a proper cluster iff they share this id
    (reference: src/sc_supertree/scs.py:508-510, 569-579).
    """
    total = tables.n_leaves
    starts = np.zeros(total, dtype=np.int64)
    # a new side starts at the first leaf of a tree and after every depth-0 gap
    gap_is_root = tables.adj_depth == 0
    starts[1:] = gap_is_root[:-1]
    starts[tables.tree_off[:-1]] = 1
    return np.cumsum(starts) - 1


def taxa_occurrences(tables: TreeTables) -> np.ndarray:
    """occ[x] = number of source trees holding taxon x.

    reference: src/sc_supertree/scs.py:580-581
    """
    return np.bincount(tables.leaf_taxon, minlength=tables.n_taxa).astype(np.int64)


def pcg_components(tables: TreeTables) -> np.ndarray:
    """Connected-component label (0..k-1, by smallest member) of every taxon (reference:
    scs.py:458-492): union-find over the root sides in ``libscs_host.so``;
    ``pcg_components_numpy`` is the vectorised reference implementation the tests compare with."""
    import ctypes as C

    from spectralclustersupertree_amd._hostlib import load

    n = tables.n_taxa
    labels = np.zeros(max(n, 1), dtype=np.int32)
    if n:
        lp, ip = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        rc = load().scs_host_components(
            n, tables.n_trees, tables.tree_off.ctypes.data_as(lp), tables.leaf_taxon.ctypes.data_as(ip),
            tables.adj_depth.ctypes.data_as(ip), labels.ctypes.data_as(ip))
        if rc:
            raise MemoryError("scs_host_components")
    return labels[:n]


def pcg_components_numpy(tables: TreeTables) -> np.ndarray:
    """Connected-component label (0..k-1, by smallest member) of every taxon.

    Edges of the proper cluster graph exist wherever the co-occurrence count is
    >= 1, independent of the edge weight
    (reference: src/sc_supertree/scs.py:458-492, 651-652), so the components
    are those of "shares a root side in some tree": union the leaves of every
    side.  Union-find with path halving, O(L alpha).
    """
    n = tables.n_taxa
    side = leaf_side_ids(tables)
    taxon = tables.leaf_taxon.astype(np.int64)
    parent = np.arange(n, dtype=np.int64)
    # first leaf of every side is the side's representative
    order = np.argsort(side, kind="stable")
    side_sorted = side[order]
    tax_sorted = taxon[order]
    first = np.ones(len(order), dtype=bool)
    first[1:] = side_sorted[1:] != side_sorted[:-1]
    rep = tax_sorted[np.maximum.accumulate(np.where(first, np.arange(len(order)), 0))]
    # iterate pointer-jumping unions until stable (vectorised label propagation)
    a, b = tax_sorted, rep
    while True:
        ra = _find_all(parent, a)
        rb = _find_all(parent, b)
        lo = np.minimum(ra, rb)
        hi = np.maximum(ra, rb)
        changed = lo != hi
        if not changed.any():
            break
        # hook larger root under smaller; duplicates resolve to the minimum
        np.minimum.at(parent, hi[changed], lo[changed])
    root = _find_all(parent, np.arange(n, dtype=np.int64))
    _, labels = np.unique(root, return_inverse=True)
    return labels.astype(np.int32)


def _find_all(parent: np.ndarray, x: np.ndarray) -> np.ndarray:
    r = parent[x]
    while True:
        nxt = parent[r]
        if np.array_equal(nxt, r):
            return r
        r = nxt


def contraction_groups(tables: TreeTables) -> np.ndarray:
    """Group id (0..V'-1, by smallest member) of every taxon after contraction: the classes of
    identical (tree, root side) signatures (see ``contraction_groups_numpy`` for why these are
    the reference's groups).  Runs in ``libscs_host.so`` (hash-based partition refinement,
    O(n_taxa) per tree); ``tests/test_treearrays.py`` checks it against the numpy version."""
    import ctypes as C

    from spectralclustersupertree_amd._hostlib import load

    n = tables.n_taxa
    groups = np.zeros(max(n, 1), dtype=np.int32)
    if n and tables.n_trees:
        lp, ip = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        rc = load().scs_host_contraction_groups(
            n, tables.n_trees, tables.tree_off.ctypes.data_as(lp), tables.leaf_taxon.ctypes.data_as(ip),
            tables.adj_depth.ctypes.data_as(ip), groups.ctypes.data_as(ip))
        if rc:
            raise MemoryError("scs_host_contraction_groups")
    return groups[:n]


def contraction_groups_numpy(tables: TreeTables) -> np.ndarray:
    """Group id (0..V'-1, by smallest member) of every taxon after contraction.

    The reference merges u, v when their co-occurrence count as a proper
    cluster equals ``max(occ[u], occ[v])`` and takes connected components of
    that relation (reference: src/sc_supertree/scs.py:302-316).  Because
    co_occ(u, v) <= min(occ[u], occ[v]), the condition holds iff u and v occur
    in exactly the same trees and on the same root side in each of them -- an
    equivalence relation -- so the groups are the classes of identical
    (tree, side) signatures.  Computed exactly (no hashing) by iterated
    refinement, one source tree at a time over the leaf table.
    """
    n = tables.n_taxa
    side = leaf_side_ids(tables)
    cls = np.zeros(n, dtype=np.int64)
    n_cls = 1
    off = tables.tree_off
    for t in range(tables.n_trees):
        lo, hi = int(off[t]), int(off[t + 1])
        tax = tables.leaf_taxon[lo:hi]
        # side index within the tree, 1-based; 0 = taxon absent from the tree
        here = np.zeros(n, dtype=np.int64)
        here[tax] = side[lo:hi] - side[lo] + 1
        n_sides = int(here.max()) + 1
        key = cls * n_sides + here
        _, cls = np.unique(key, return_inverse=True)
        n_cls = int(cls.max()) + 1
        if n_cls == n:
            break
    # relabel by smallest member
    first = np.full(n_cls, n, dtype=np.int64)
    np.minimum.at(first, cls, np.arange(n, dtype=np.int64))
    rank = np.argsort(np.argsort(first))
    return rank[cls].astype(np.int32)
