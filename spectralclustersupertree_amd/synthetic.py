"""Deterministic synthetic source trees via libscs_synth.so.

Host-only helper for benchmarks and parity tests: iid random-join rooted binary
trees over ``n_taxa`` taxa (optionally only ``leaves_per_tree`` of them per
tree), Exp(mean 0.1) branch lengths, supports in 50..100, emitted directly as
flattened tables.  ``tree_objects`` rebuilds the same trees as ``TreeNode``
objects so the dict-based oracle can be fed the identical input.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from spectralclustersupertree_amd.flatten import STRATEGIES, TreeTables
from spectralclustersupertree_amd.tree import TreeNode

_LIB_PATH = Path(__file__).resolve().parent / "libscs_synth.so"
_lib = None


def _load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            msg = f"{_LIB_PATH} not found: run __graft_entry__.build()"
            raise ImportError(msg)
        lib = C.CDLL(str(_LIB_PATH))
        ip, dp, lp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
        lib.scs_synth_tree.restype = C.c_int
        lib.scs_synth_tree.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                       ip, ip, dp, ip, ip, dp, dp, ip]
        lib.scs_synth_tree_nodes.restype = C.c_int
        lib.scs_synth_tree_nodes.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.scs_synth_tables.restype = C.c_int
        lib.scs_synth_tables.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, lp, ip, ip, dp, dp]
        lib.scs_synth_tables_planted.restype = C.c_int
        lib.scs_synth_tables_planted.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_int32, lp, ip, ip, dp, dp]
        _lib = lib
    return _lib


def taxon_name(i: int) -> str:
    """Zero-padded so that lexicographic name order equals id order."""
    return f"t{i:07d}"


def make_tables(
    seed: int,
    n_taxa: int,
    n_trees: int,
    strategy: str,
    leaves_per_tree: int | None = None,
    random_weights: bool = False,
    planted_spr: int | None = None,
    pinned: bool = False,
) -> TreeTables:
    """Flattened tables of the synthetic set ``(seed, n_taxa, n_trees)``.

    ``planted_spr`` (the planted variant): every tree is the set's
    model tree over all taxa plus that many random SPR moves, instead of an independent
    random-join tree.  ``pinned``: the arrays live in page-locked host memory
    (``_native.pinned_empty``; needs a HIP device) so that ``Device.upload`` moves them by DMA.
    """
    lib = _load()
    k = n_taxa if leaves_per_tree is None else int(leaves_per_tree)
    if planted_spr is not None and k != n_taxa:
        msg = "planted sets cover all taxa in every tree"
        raise ValueError(msg)
    total = n_trees * k
    empty = np.empty
    if pinned:
        from spectralclustersupertree_amd._native import pinned_empty as empty
    tree_off = empty(n_trees + 1, dtype=np.int64)
    leaf_taxon = empty(total, dtype=np.int32)
    adj_depth = empty(total, dtype=np.int32)
    adj_val = empty(total, dtype=np.float64)
    tree_w = empty(n_trees, dtype=np.float64)
    ip, dp, lp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    if planted_spr is None:
        rc = lib.scs_synth_tables(
            seed, n_taxa, n_trees, k, STRATEGIES_INDEX[strategy], 1 if random_weights else 0,
            tree_off.ctypes.data_as(lp), leaf_taxon.ctypes.data_as(ip), adj_depth.ctypes.data_as(ip),
            adj_val.ctypes.data_as(dp), tree_w.ctypes.data_as(dp),
        )
    else:
        rc = lib.scs_synth_tables_planted(
            seed, n_taxa, n_trees, STRATEGIES_INDEX[strategy], 1 if random_weights else 0,
            int(planted_spr), tree_off.ctypes.data_as(lp), leaf_taxon.ctypes.data_as(ip),
            adj_depth.ctypes.data_as(ip), adj_val.ctypes.data_as(dp), tree_w.ctypes.data_as(dp),
        )
    if rc != 0:
        msg = "scs_synth_tables failed (bad arguments or out of memory)"
        raise RuntimeError(msg)
    return TreeTables(n_taxa, tree_off, leaf_taxon, adj_depth, adj_val, tree_w,
                      [taxon_name(i) for i in range(n_taxa)] if n_taxa <= 200000 else None,
                      # Exp() branch lengths are positive: everything but bootstrap is monotone
                      monotone=strategy != "bootstrap")


STRATEGIES_INDEX = {"one": 0, "depth": 1, "branch": 2, "bootstrap": 3}
assert set(STRATEGIES_INDEX) == set(STRATEGIES)


def tree_objects(
    seed: int, n_taxa: int, n_trees: int, leaves_per_tree: int | None = None
) -> list[TreeNode]:
    """The same trees as ``make_tables`` as ``TreeNode`` objects (small sizes)."""
    lib = _load()
    k = n_taxa if leaves_per_tree is None else int(leaves_per_tree)
    nn = 2 * k - 1
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    out = []
    for t in range(n_trees):
        lt = np.empty(k, dtype=np.int32)
        ad = np.empty(k, dtype=np.int32)
        av = np.empty(k, dtype=np.float64)
        left = np.empty(nn, dtype=np.int32)
        right = np.empty(nn, dtype=np.int32)
        length = np.empty(nn, dtype=np.float64)
        support = np.empty(nn, dtype=np.float64)
        tax = np.empty(k, dtype=np.int32)
        rc = lib.scs_synth_tree(
            seed, t, n_taxa, k, 0, lt.ctypes.data_as(ip), ad.ctypes.data_as(ip),
            av.ctypes.data_as(dp), left.ctypes.data_as(ip), right.ctypes.data_as(ip),
            length.ctypes.data_as(dp), support.ctypes.data_as(dp), tax.ctypes.data_as(ip),
        )
        if rc != 0:
            msg = "scs_synth_tree failed"
            raise RuntimeError(msg)
        nodes = [TreeNode(taxon_name(int(tax[v])), None, float(length[v]), None) for v in range(k)]
        for v in range(k, nn):
            node = TreeNode(None, None, float(length[v]), float(support[v]))
            node.append(nodes[int(left[v])])
            node.append(nodes[int(right[v])])
            nodes.append(node)
        root = nodes[-1]
        root.length = None
        out.append(root)
    return out


def tree_arrays(seed: int, n_taxa: int, n_trees: int, leaves_per_tree: int | None = None,
                random_weights: bool = False):
    """The same trees as ``tree_objects`` / ``make_tables`` as flat node arrays
    (``treearrays.TreeArrays``), built in C without tree objects: the input of whole-recursion
    runs at sizes where a Python object per node is out of the question."""
    from .treearrays import TreeArrays

    lib = _load()
    k = n_taxa if leaves_per_tree is None else int(leaves_per_tree)
    nn = 2 * k - 1
    parent = np.empty(n_trees * nn, dtype=np.int32)
    taxon = np.empty(n_trees * nn, dtype=np.int32)
    length = np.empty(n_trees * nn, dtype=np.float64)
    support = np.empty(n_trees * nn, dtype=np.float64)
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    for t in range(n_trees):
        sl = slice(t * nn, (t + 1) * nn)
        rc = lib.scs_synth_tree_nodes(seed, t, n_taxa, k, parent[sl].ctypes.data_as(ip),
                                      taxon[sl].ctypes.data_as(ip), length[sl].ctypes.data_as(dp),
                                      support[sl].ctypes.data_as(dp))
        if rc != 0:
            msg = "scs_synth_tree_nodes failed"
            raise RuntimeError(msg)
    weights = np.ones(n_trees, dtype=np.float64)
    if random_weights:
        # (a tree's weight depends on (seed, tree) only: take it from a two-leaf table set)
        weights = make_tables(seed, n_taxa, n_trees, "one", leaves_per_tree=min(2, n_taxa),
                              random_weights=True).tree_w.copy()
    return TreeArrays(n_taxa=n_taxa, node_off=np.arange(n_trees + 1, dtype=np.int64) * nn, parent=parent,
                      taxon=taxon, length=length, support=support, weights=weights,
                      taxa=[taxon_name(i) for i in range(n_taxa)])
