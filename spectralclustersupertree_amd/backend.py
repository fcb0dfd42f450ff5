"""Host-side handles on the device objects of libscs_hip.so.

``Device`` owns one context (one process <-> one MI355X); ``DeviceGraph`` is a
row block of the proper-cluster-graph weight matrix resident in HBM.  These are
thin: every number is produced by the HIP kernels behind include/scs_hip.h.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from spectralclustersupertree_amd import _env
from spectralclustersupertree_amd import _native as nv
from spectralclustersupertree_amd.flatten import TreeTables

DEFAULT_TOL = 1e-13
DEFAULT_MAX_ITER = 2000


def _resident_solve() -> bool:
    """SCS_RESIDENT_SOLVE=0 (diagnostic): small nodes pack their tables on the host even when the
    tables are resident on the device."""
    return bool(int(_env.probe("SCS_RESIDENT_SOLVE", "1")))


class Device:
    """One libscs_hip context.  ``Device()`` = GPU 0, single rank."""

    def __init__(self, device: int = 0, rank: int = 0, world: int = 1, unique_id: bytes | None = None,
                 *, _local_group=None) -> None:
        self._lib = nv.load_library()
        self._ctx = C.c_void_p()
        self.rank, self.world = rank, world
        self.index = device  # the GPU this context lives on
        if _local_group is not None:
            nv.check(self._lib.scs_ctx_create_local(device, rank, _local_group, C.byref(self._ctx)))
        else:
            uid = None
            if unique_id is not None:
                if len(unique_id) != nv.UNIQUE_ID_BYTES:
                    msg = f"unique_id must be {nv.UNIQUE_ID_BYTES} bytes"
                    raise ValueError(msg)
                uid = C.create_string_buffer(bytes(unique_id), nv.UNIQUE_ID_BYTES)
            nv.check(self._lib.scs_ctx_create(device, rank, world, uid, C.byref(self._ctx)))

    @staticmethod
    def unique_id() -> bytes:
        """RCCL bootstrap id; rank 0 creates it and ships it to the other ranks."""
        lib = nv.load_library()
        buf = C.create_string_buffer(nv.UNIQUE_ID_BYTES)
        nv.check(lib.scs_comm_unique_id(buf))
        return buf.raw

    def close(self) -> None:
        if self._ctx:
            self._lib.scs_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self) -> "Device":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def synchronize(self) -> None:
        nv.check(self._lib.scs_ctx_synchronize(self._ctx))

    def trim(self, keep_bytes: int = 0) -> None:
        """``scs_ctx_trim``: hand whole free slabs of the device's arena back to the driver until at most
        ``keep_bytes`` of free arena memory remain (the arena is shared by every context of the process on this
        device and never shrinks by itself)."""
        nv.check(self._lib.scs_ctx_trim(self._ctx, int(keep_bytes)))

    def reserve(self, n_bytes: int) -> None:
        """``scs_ctx_reserve``: the device's arena holds ``n_bytes`` of free memory in one piece afterwards."""
        nv.check(self._lib.scs_ctx_reserve(self._ctx, int(n_bytes)))

    def arena_stats(self) -> dict:
        """``scs_debug_arena_stats`` of this context's device."""
        out = (C.c_int64 * 8)()
        nv.check(self._lib.scs_debug_arena_stats(int(self.index), out))
        keys = ("slab_bytes", "used_bytes", "slabs", "chunks", "pending_chunks", "driver_allocations",
                "driver_releases", "requests")
        return dict(zip(keys, (int(x) for x in out)))

    def comm_info(self) -> dict:
        """The communicator as it sees itself (``scs_ctx_comm_info``): kind, the world / rank it was
        created with and what ncclCommCount / ncclCommUserRank report (-1: not available)."""
        v = [C.c_int32(0) for _ in range(5)]
        nv.check(self._lib.scs_ctx_comm_info(self._ctx, *[C.byref(x) for x in v]))
        kind, world, rank, rep_world, rep_rank = (int(x.value) for x in v)
        return {"kind": {0: "none", 1: "rccl", 2: "in-process team"}.get(kind, str(kind)), "world": world,
                "rank": rank, "reported_world": rep_world, "reported_rank": rep_rank}

    # -- tables ------------------------------------------------------------
    def upload(self, tables: TreeTables) -> "DeviceTables":
        tables.validate(ranges=False)  # (ranges: checked by a kernel on the uploaded copy)
        handle = C.c_void_p()
        res = getattr(tables, "resident", None)
        if res is not None and _resident_solve() and res[0].dev.index == self.index and res[0]._h:
            # the tables are on this device already (a child of scs_forest_split): device-to-device
            forest, relabel = res
            rl = None if relabel is None else np.ascontiguousarray(relabel, dtype=np.int32)
            nv.check(self._lib.scs_tables_from_forest(self._ctx, forest._h, nv.iptr(rl) if rl is not None else None,
                                                      int(tables.n_taxa), C.byref(handle)))
            return DeviceTables(self, handle, tables.n_taxa, tables.n_trees, bool(tables.monotone))
        rc = self._lib.scs_tables_upload(
            self._ctx, tables.n_taxa, tables.n_trees, nv.lptr(tables.tree_off),
            nv.iptr(tables.leaf_taxon), nv.iptr(tables.adj_depth), nv.dptr(tables.adj_val),
            nv.dptr(tables.tree_w), C.byref(handle),
        )
        if rc == nv.EINVAL:
            msg = self._lib.scs_last_error()
            raise ValueError(msg.decode() if msg else "scs_tables_upload: invalid tables")
        nv.check(rc)
        out = DeviceTables(self, handle, tables.n_taxa, tables.n_trees, bool(tables.monotone))
        out._source = tables  # (page-locked arrays may still be read until the first build returns)
        return out

    # -- batched small nodes --------------------------------------------------
    # largest node of the batched path (SMALL_MAXS of libscs_hip: two-sided Jacobi in LDS up to 64
    # vertices, one-sided up to 128 -- SURVEY.md 8f rank 3).  SCS_SMALL_MAX_TAXA moves the limit down.
    SMALL_MAX_TAXA = max(2, min(128, int(_env.probe("SCS_SMALL_MAX_TAXA", "128"))))

    def small_solve(self, nodes, want_w: bool = False):
        """K small recursion nodes in one launch (``scs_small_solve``; reference: scs.py:110-134
        at depth).  ``nodes``: list of ``(tables, group_start or None)`` with at most
        ``SMALL_MAX_TAXA`` taxa each, taxa numbered so that contraction groups are consecutive
        ranges.  Returns one ``(maps, lambdas)`` -- or ``(maps, lambdas, W)`` -- per node."""
        if len(nodes) == 0:
            return []
        return self.small_solve_begin(nodes, want_w).result()

    def small_solve_begin(self, nodes, want_w: bool = False) -> "SmallTicket":
        """The same, not waited for (``scs_small_solve_begin``): the returned ticket's
        ``result()`` is the list ``small_solve`` returns.  At least one node."""
        k = len(nodes)
        if k == 1 and getattr(nodes[0][0], "resident", None) is not None and _resident_solve():
            forest, relabel = nodes[0][0].resident
            if forest.dev is self and forest._h:
                # the node's tables are on this device already (a child of scs_forest_split): only the
                # group boundaries and the renumbering travel (scs_small_solve_begin_forest)
                tables, group_start = nodes[0]
                gs = (np.arange(tables.n_taxa + 1, dtype=np.int32) if group_start is None
                      else np.ascontiguousarray(group_start, dtype=np.int32))
                rl = None if relabel is None else np.ascontiguousarray(relabel, dtype=np.int32)
                ticket = C.c_int32(-1)
                nv.check(self._lib.scs_small_solve_begin_forest(
                    self._ctx, forest._h, nv.iptr(rl) if rl is not None else None, int(tables.n_taxa), len(gs) - 1,
                    nv.iptr(gs), int(want_w), C.byref(ticket)))
                return SmallTicket(self, ticket.value, np.asarray([len(gs) - 1], dtype=np.int32), want_w, list(nodes))
        n_taxa = np.empty(k, dtype=np.int32)
        n_trees = np.empty(k, dtype=np.int32)
        n_groups = np.empty(k, dtype=np.int32)
        toff, gstart, lt, ad, av, tw = [], [], [], [], [], []
        for i, (tables, group_start) in enumerate(nodes):
            n_taxa[i], n_trees[i] = tables.n_taxa, tables.n_trees
            gs = (np.arange(tables.n_taxa + 1, dtype=np.int32) if group_start is None
                  else np.ascontiguousarray(group_start, dtype=np.int32))
            n_groups[i] = len(gs) - 1
            toff.append(np.asarray(tables.tree_off, dtype=np.int32))
            gstart.append(gs)
            lt.append(tables.leaf_taxon)
            ad.append(tables.adj_depth)
            av.append(tables.adj_val)
            tw.append(tables.tree_w)
        if k == 1:
            cat = lambda parts: parts[0]  # noqa: E731
        else:
            cat = np.concatenate
        toff_a = np.ascontiguousarray(cat(toff), dtype=np.int32)
        gs_a = np.ascontiguousarray(cat(gstart), dtype=np.int32)
        lt_a = np.ascontiguousarray(cat(lt), dtype=np.int32)
        ad_a = np.ascontiguousarray(cat(ad), dtype=np.int32)
        av_a = np.ascontiguousarray(cat(av), dtype=np.float64)
        tw_a = np.ascontiguousarray(cat(tw), dtype=np.float64)
        ticket = C.c_int32(-1)
        nv.check(self._lib.scs_small_solve_begin(
            self._ctx, k, nv.iptr(n_taxa), nv.iptr(n_trees), nv.iptr(n_groups), nv.iptr(toff_a),
            nv.iptr(lt_a), nv.iptr(ad_a), nv.dptr(av_a), nv.dptr(tw_a), nv.iptr(gs_a), int(want_w),
            C.byref(ticket)))
        return SmallTicket(self, ticket.value, n_groups, want_w, list(nodes))

    def small_solve_begin_level(self, forest: "DeviceForest", t_begin, n_trees, n_leaves, u_base, u_size, relabel,
                                n_taxa, n_groups, group_start) -> "SmallTicket":
        """``scs_small_solve_begin_level``: the K small nodes of one level of the recursion from the level
        forest's resident tables (``levels.Engine``; reference: scs.py:110-134 for every node of the level).
        All arguments are arrays over the K nodes (``relabel`` / ``group_start`` concatenated)."""
        k = len(t_begin)
        as32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        t_begin, n_trees, u_base, u_size = as32(t_begin), as32(n_trees), as32(u_base), as32(u_size)
        relabel, n_taxa, n_groups, group_start = as32(relabel), as32(n_taxa), as32(n_groups), as32(group_start)
        n_leaves = np.ascontiguousarray(n_leaves, dtype=np.int64)
        ticket = C.c_int32(-1)
        nv.check(self._lib.scs_small_solve_begin_level(
            self._ctx, forest._h, k, nv.iptr(t_begin), nv.iptr(n_trees), nv.lptr(n_leaves), nv.iptr(u_base),
            nv.iptr(u_size), nv.iptr(relabel), nv.iptr(n_taxa), nv.iptr(n_groups), nv.iptr(group_start), 0,
            C.byref(ticket)))
        return SmallTicket(self, ticket.value, n_groups, False, None)

    def upload_range(self, forest: "DeviceForest", t_begin: int, t_end: int, u_base: int, u_size: int,
                     relabel: np.ndarray, n_taxa: int, monotone: bool) -> "DeviceTables":
        """``scs_tables_from_forest_range``: the tables of ONE node of a level forest as a handle for
        ``build`` (the nodes of a level that are too large for the batched small path)."""
        handle = C.c_void_p()
        rl = np.ascontiguousarray(relabel, dtype=np.int32)
        nv.check(self._lib.scs_tables_from_forest_range(self._ctx, forest._h, int(t_begin), int(t_end), int(u_base),
                                                        int(u_size), nv.iptr(rl), int(n_taxa), C.byref(handle)))
        return DeviceTables(self, handle, int(n_taxa), int(t_end - t_begin), bool(monotone))

    def copy_bandwidth(self, nbytes: int = 1 << 30, reps: int = 6) -> float:
        """Measured device-to-device copy rate of this GPU in GB/s (read + write bytes over
        time; ``scs_debug_copy_bandwidth``) -- the figure to hold beside the nominal HBM peak."""
        out = C.c_double(0.0)
        nv.check(self._lib.scs_debug_copy_bandwidth(self._ctx, int(nbytes), int(reps), C.byref(out)))
        return float(out.value)

    # -- building blocks exposed for the parity tests -----------------------
    def comm_selftest(self, x: np.ndarray) -> np.ndarray:
        """One grouped ncclSend/ncclRecv round to this rank itself plus an ncclAllGather,
        through the library's RCCL wrappers (``scs_debug_comm_selftest``); returns what came
        back.  The context must own an RCCL communicator (created with a unique id)."""
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        out = np.empty_like(x)
        nv.check(self._lib.scs_debug_comm_selftest(self._ctx, len(x), nv.dptr(x), nv.dptr(out)))
        return out


    def debug_jacobi(self, a: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float64)
        n = a.shape[0]
        w = np.empty(n)
        v = np.empty((n, n))
        nv.check(self._lib.scs_debug_jacobi(self._ctx, nv.dptr(a), n, nv.dptr(w), nv.dptr(v)))
        return w, v

    def debug_gram(self, a: np.ndarray, b: np.ndarray, use_mfma: bool):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty((a.shape[1], b.shape[1]))
        nv.check(self._lib.scs_debug_gram(self._ctx, nv.dptr(a), nv.dptr(b), a.shape[0], a.shape[1],
                                          b.shape[1], int(use_mfma), nv.dptr(out)))
        return out


class DeviceForest:
    """A source forest resident in HBM (``scs_forest``): preorder node arrays, and -- for the
    children of ``split`` -- their flattened tables.  See ``treearrays.ResidentArrays``."""

    def __init__(self, dev: Device, handle, n_taxa: int, n_trees: int, n_nodes: int, n_leaves: int,
                 monotone_flag: bool = True) -> None:
        self.dev, self._h = dev, handle
        self.n_taxa, self.n_trees, self.n_nodes, self.n_leaves = n_taxa, n_trees, n_nodes, n_leaves
        self.monotone_flag = monotone_flag

    @classmethod
    def upload(cls, dev: Device, n_taxa, node_off, parent, taxon, length, support, weights, n_leaves) -> "DeviceForest":
        handle = C.c_void_p()
        nv.check(dev._lib.scs_forest_upload(dev._ctx, int(n_taxa), len(weights), nv.lptr(node_off), nv.iptr(parent),
                                            nv.iptr(taxon), nv.dptr(length), nv.dptr(support), nv.dptr(weights),
                                            int(n_leaves), C.byref(handle)))
        return cls(dev, handle, int(n_taxa), len(weights), int(node_off[-1]), int(n_leaves))

    def free(self) -> None:
        if self._h:
            self.dev._lib.scs_forest_free(self.dev._ctx, self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            if self.dev._ctx:
                self.free()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def split(self, part_of: np.ndarray, new_id: np.ndarray, part_taxa, strategy_code: int) -> list["DeviceForest"]:
        """``scs_forest_split``: one child per part, each with its tables (reference: scs.py:411-455)."""
        n_parts = len(part_taxa)
        handles = (C.c_void_p * n_parts)()
        info = (nv.ForestInfo * n_parts)()
        pt = np.ascontiguousarray(part_taxa, dtype=np.int32)
        rc = self.dev._lib.scs_forest_split(self.dev._ctx, self._h, nv.iptr(part_of), nv.iptr(new_id), n_parts,
                                            nv.iptr(pt), int(strategy_code), handles, info)
        if rc == nv.EUNSUP:
            # the reference fails in ``length * tree_weight`` with a missing support (scs.py:656)
            msg = "unsupported operand type(s) for *: 'NoneType' and 'float'"
            raise TypeError(msg)
        nv.check(rc)
        return [DeviceForest(self.dev, C.c_void_p(handles[c]), int(pt[c]), int(info[c].n_trees), int(info[c].n_nodes),
                             int(info[c].n_leaves), bool(info[c].monotone)) for c in range(n_parts)]

    def split_level(self, part_of: np.ndarray, new_id: np.ndarray, n_parts: int, child_taxa: int, strategy_code: int,
                    node_tree_end: np.ndarray):
        """``scs_forest_split_level``: every node of a level (consecutive tree ranges of this forest, ``node_tree_end``)
        restricted to every one of its parts in ONE call (reference: scs.py:411-455 for every node of the level).
        Returns ``(union forest, child_trees [n_parts, K], child_leaves [n_parts, K], present, comp_root, sig)``."""
        k = len(node_tree_end)
        handle = C.c_void_p()
        info = nv.ForestInfo()
        nte = np.ascontiguousarray(node_tree_end, dtype=np.int32)
        child_trees = np.zeros((n_parts, k), dtype=np.int32)
        child_leaves = np.zeros((n_parts, k), dtype=np.int64)
        present = np.zeros(child_taxa, dtype=np.uint8)
        comp_root = np.zeros(child_taxa, dtype=np.int32)
        sig = np.zeros((child_taxa, 2), dtype=np.uint64)
        rc = self.dev._lib.scs_forest_split_level(
            self.dev._ctx, self._h, nv.iptr(part_of), nv.iptr(new_id), int(n_parts), int(child_taxa),
            int(strategy_code), k, nv.iptr(nte), C.byref(handle), C.byref(info), nv.iptr(child_trees),
            nv.lptr(child_leaves), present.ctypes.data, nv.iptr(comp_root), sig.ctypes.data)
        if rc == nv.EUNSUP:
            # the reference fails in ``length * tree_weight`` with a missing support (scs.py:656)
            msg = "unsupported operand type(s) for *: 'NoneType' and 'float'"
            raise TypeError(msg)
        nv.check(rc)
        union = DeviceForest(self.dev, handle, int(child_taxa), int(info.n_trees), int(info.n_nodes),
                             int(info.n_leaves), bool(info.monotone))
        return union, child_trees, child_leaves, present, comp_root, sig

    def slice(self, t_begin: int, t_end: int) -> "DeviceForest":
        """``scs_forest_slice``: the trees [t_begin, t_end) as a forest of their own, nothing copied (taxon ids stay
        this forest's)."""
        handle = C.c_void_p()
        nv.check(self.dev._lib.scs_forest_slice(self.dev._ctx, self._h, int(t_begin), int(t_end), C.byref(handle)))
        out = DeviceForest(self.dev, handle, self.n_taxa, int(t_end - t_begin), 0, 0, self.monotone_flag)
        out._base = self  # (the slice points into this forest's arrays)
        return out

    def analyze(self):
        """``scs_forest_analyze``: ``(comp_root [n_taxa], sig [n_taxa, 2])`` of a forest that carries tables."""
        comp_root = np.zeros(max(self.n_taxa, 1), dtype=np.int32)
        sig = np.zeros((max(self.n_taxa, 1), 2), dtype=np.uint64)
        nv.check(self.dev._lib.scs_forest_analyze(self.dev._ctx, self._h, nv.iptr(comp_root), sig.ctypes.data))
        return comp_root, sig

    def tables_range(self, t_begin: int, t_end: int):
        """``(tree_off, leaf_taxon, adj_depth, adj_val, tree_w)`` of the trees [t_begin, t_end) (host copies)."""
        m = int(t_end - t_begin)
        tree_off = np.zeros(m + 1, dtype=np.int64)
        nv.check(self.dev._lib.scs_forest_tables_download_range(self.dev._ctx, self._h, int(t_begin), int(t_end),
                                                                nv.lptr(tree_off), None, None, None, None))
        l = int(tree_off[-1])
        leaf_taxon = np.empty(l, dtype=np.int32)
        adj_depth = np.empty(l, dtype=np.int32)
        adj_val = np.empty(l, dtype=np.float64)
        tree_w = np.empty(m, dtype=np.float64)
        nv.check(self.dev._lib.scs_forest_tables_download_range(
            self.dev._ctx, self._h, int(t_begin), int(t_end), nv.lptr(tree_off), nv.iptr(leaf_taxon) if l else None,
            nv.iptr(adj_depth) if l else None, nv.dptr(adj_val) if l else None, nv.dptr(tree_w) if m else None))
        return tree_off, leaf_taxon, adj_depth, adj_val, tree_w

    def tables(self):
        """``(tree_off, leaf_taxon, adj_depth, adj_val, tree_index, tree_w, present)`` of a child of
        ``split``: read-only views of the page-locked host copy the split left (no further copy; the
        views keep this forest -- and with it the block -- alive)."""
        m, l = self.n_trees, self.n_leaves
        ptrs = [C.c_void_p() for _ in range(7)]
        nv.check(self.dev._lib.scs_forest_tables_host(self.dev._ctx, self._h, *[C.byref(x) for x in ptrs]))

        def view(ptr, dtype, count):
            dt = np.dtype(dtype)
            if count == 0:
                return np.empty(0, dtype=dt)
            buf = (C.c_char * (count * dt.itemsize)).from_address(ptr.value)
            buf._scs_owner = self
            a = np.frombuffer(buf, dtype=dt, count=count)
            a.flags.writeable = False
            return a

        return (view(ptrs[0], np.int64, m + 1), view(ptrs[1], np.int32, l), view(ptrs[2], np.int32, l),
                view(ptrs[3], np.float64, l), view(ptrs[4], np.int32, m), view(ptrs[5], np.float64, m),
                view(ptrs[6], np.uint8, max(self.n_taxa, 0)))

    def download(self, t_begin: int = 0, t_end: int | None = None):
        """Node arrays of the trees ``[t_begin, t_end)``: ``(node_off, parent, taxon, length, support, weights)``."""
        t_end = self.n_trees if t_end is None else t_end
        m = t_end - t_begin
        node_off = np.zeros(m + 1, dtype=np.int64)
        # (sizes are known only after the offsets arrive: two calls)
        nv.check(self.dev._lib.scs_forest_download(self.dev._ctx, self._h, t_begin, t_end, nv.lptr(node_off),
                                                   None, None, None, None, None))
        n = int(node_off[-1])
        parent = np.empty(n, dtype=np.int32)
        taxon = np.empty(n, dtype=np.int32)
        length = np.empty(n, dtype=np.float64)
        support = np.empty(n, dtype=np.float64)
        weights = np.empty(m, dtype=np.float64)
        nv.check(self.dev._lib.scs_forest_download(self.dev._ctx, self._h, t_begin, t_end, nv.lptr(node_off),
                                                   nv.iptr(parent) if n else None, nv.iptr(taxon) if n else None,
                                                   nv.dptr(length) if n else None, nv.dptr(support) if n else None,
                                                   nv.dptr(weights) if m else None))
        return node_off, parent, taxon, length, support, weights


# A module attribute only a test sets (never the environment: ADVICE r05): the batched path then reports the nodes of
# more than 64 vertices as unconverged, and the per-node general path below takes them.
_FAIL_SMALL_FOR_TESTS = False


class SmallTicket:
    """An ``scs_small_solve_begin`` that has not been ended: ``result()`` waits and returns one
    ``(maps, lambdas[, W])`` per node (once; kept).  Dropped unasked-for, it releases its slot."""

    def __init__(self, dev: Device, ticket: int, n_groups: np.ndarray, want_w: bool, nodes=None) -> None:
        self.dev, self._ticket, self._n_groups, self._want_w = dev, ticket, n_groups, want_w
        self._nodes = nodes
        self._out = None

    def _general_path(self, i: int):
        """Node ``i`` once more through the general per-node path (upload, build, contract, LOBPCG
        with an explicit block width -- which keeps scs_fiedler off the dense one-sided solve that
        has just failed): ``(maps, lambdas, W or None)``.  Raises RuntimeError if that fails too."""
        tables, group_start = self._nodes[i]
        dtab = self.dev.upload(tables)
        try:
            graph = dtab.build()
        finally:
            dtab.free()
        try:
            if group_start is not None:
                graph = graph.contract(group_start)
            last = None
            # (a graph too small for an explicit block width -- a 65..128-taxon node may contract to a handful of
            # vertices -- takes the library's default path: ADVICE r05)
            widths = [(b, it) for b, it in ((4, DEFAULT_MAX_ITER), (8, 4 * DEFAULT_MAX_ITER)) if graph.shape[0] > 3 * b + 2]
            for block, iters in widths or [(0, 4 * DEFAULT_MAX_ITER)]:
                try:
                    maps, stats = graph.fiedler(None, block=block, max_iter=iters)
                    lam = np.array([stats["lambda"][0], stats["lambda"][1], stats["lambda_next"]])
                    return maps, lam, (graph.download() if self._want_w else None)
                except nv.ConvergenceError as exc:
                    last = exc
            msg = f"small-node eigen-solve did not converge, nor did the general path ({last})"
            raise RuntimeError(msg)
        finally:
            graph.free()

    def raw(self):
        """Waits and returns ``(maps [sum n_groups, 2], lambdas [K, 3])`` as the library left them (a node whose
        one-sided Jacobi ran out of sweeps carries NaN eigenvalues: the caller decides)."""
        n_groups = self._n_groups
        maps = np.empty((int(n_groups.sum()), 2))
        lam = np.empty((len(n_groups), 3))
        ticket, self._ticket = self._ticket, -1
        nv.check(self.dev._lib.scs_small_solve_end(self.dev._ctx, ticket, nv.dptr(maps), nv.dptr(lam), None))
        return maps, lam

    def result(self):
        if self._out is None:
            n_groups, want_w = self._n_groups, self._want_w
            k = len(n_groups)
            maps = np.empty((int(n_groups.sum()), 2))
            lam = np.empty((k, 3))
            w = np.empty(int((n_groups.astype(np.int64) ** 2).sum())) if want_w else None
            ticket, self._ticket = self._ticket, -1
            nv.check(self.dev._lib.scs_small_solve_end(self.dev._ctx, ticket, nv.dptr(maps), nv.dptr(lam),
                                                       nv.dptr(w) if want_w else None))
            redo = {}
            if _FAIL_SMALL_FOR_TESTS:  # (tests/test_gpu_parity.py monkeypatches it: the one-sided Jacobi "gave up")
                lam[n_groups > 64] = np.nan
            if not np.all(np.isfinite(lam)):
                # the one-sided Jacobi of a node of more than 64 vertices ran out of sweeps (NaN
                # eigenvalues, never a half-rotated basis): only THAT node goes through the general
                # path again; if that fails too the call raises, as the reference's ARPACK call
                # raises rather than return a guess (scs.py:252)
                bad = [i for i in range(k) if not np.all(np.isfinite(lam[i]))]
                if self._nodes is None:
                    msg = f"small-node eigen-solve did not converge (nodes {bad} of a batch of {k})"
                    raise RuntimeError(msg)
                for i in bad:
                    redo[i] = self._general_path(i)
            out, at, wat = [], 0, 0
            for i in range(k):
                v = int(n_groups[i])
                if i in redo:
                    item = (redo[i][0], redo[i][1]) + ((redo[i][2],) if want_w else ())
                else:
                    item = (maps[at:at + v], lam[i])
                    if want_w:
                        item += (w[wat:wat + v * v].reshape(v, v),)
                if want_w:
                    wat += v * v
                out.append(item)
                at += v
            self._nodes = None
            self._out = out
        return self._out

    def __del__(self) -> None:
        try:
            if self._ticket >= 0 and self.dev._ctx:
                self.dev._lib.scs_small_solve_end(self.dev._ctx, self._ticket, None, None, None)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class DeviceTables:
    def __init__(self, dev: Device, handle, n_taxa: int, n_trees: int, monotone: bool = False) -> None:
        self.dev, self._h = dev, handle
        self.n_taxa, self.n_trees = n_taxa, n_trees
        self.monotone = monotone

    def free(self) -> None:
        if self._h:
            self.dev._lib.scs_tables_free(self.dev._ctx, self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass

    def build(self, row_begin: int = 0, row_end: int | None = None,
              shared: bool | None = None, upper: bool = False, scatter: bool = False) -> "DeviceGraph":
        """Rows [row_begin, row_end) of W (reference: scs.py:495-663).

        ``shared`` (default: on whenever the device belongs to a multi-rank job) makes the
        call collective: the ranks split the upper-triangle tiles, exchange them and keep
        their own rows, so no cell is evaluated twice across the job.

        ``upper`` (multi-rank jobs, instead of ``shared``): every rank keeps only the tiles on
        and right of the diagonal of its own rows -- no exchange, half the memory, and the
        solve streams half the bytes (``SCS_BUILD_UPPER``; ``row_begin`` a multiple of 256, see
        ``partition.row_splits_upper``).  Not for nodes that contract.
        """
        if row_end is None:
            row_end = self.n_taxa
        handle = C.c_void_p()
        stats = nv.BuildStats()
        flags = nv.BUILD_MONOTONE if self.monotone else 0
        if upper:
            flags |= nv.BUILD_UPPER
            shared = False
        if scatter:  # the atomic-scatter comparison variant (measurements only)
            flags |= nv.BUILD_SCATTER
            shared = False
        if shared is None:
            shared = self.dev.world > 1
        if shared:
            flags |= nv.BUILD_SHARED
        nv.check(self.dev._lib.scs_pcg_build(self.dev._ctx, self._h, row_begin, row_end, flags,
                                             C.byref(handle), C.byref(stats)))
        graph = DeviceGraph(self.dev, handle, stats.as_dict())
        graph.upper = bool(upper)
        return graph


    def matrix_free_graph(self, max_block: int = 8) -> "DeviceGraph":
        """A graph WITHOUT its matrix (``scs_graph_matrix_free``): ``fiedler`` on it applies W straight
        from these tables -- a measured comparison (``tools/matrix_free_compare.py``), not the product
        path; the tables must outlive the graph; no ``download`` / ``contract``."""
        handle = C.c_void_p()
        nv.check(self.dev._lib.scs_graph_matrix_free(self.dev._ctx, self._h, int(max_block), C.byref(handle)))
        return DeviceGraph(self.dev, handle, {})


class DeviceGraph:
    def __init__(self, dev: Device, handle, build_stats: dict | None = None) -> None:
        self.dev, self._h = dev, handle
        self.build_stats = build_stats or {}
        self.upper = False  # an upper-triangle job (SCS_BUILD_UPPER): block widths 4 and 8 only

    def free(self) -> None:
        if self._h:
            self.dev._lib.scs_graph_free(self.dev._ctx, self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass

    @property
    def shape(self) -> tuple[int, int, int]:
        """(V, row_begin, row_end)."""
        n, rb, re_ = C.c_int32(), C.c_int32(), C.c_int32()
        nv.check(self.dev._lib.scs_graph_shape(self._h, C.byref(n), C.byref(rb), C.byref(re_)))
        return n.value, rb.value, re_.value

    def contract(self, group_start: np.ndarray) -> "DeviceGraph":
        """Merge consecutive index ranges (reference: scs.py:336-387); consumes self."""
        gs = np.ascontiguousarray(group_start, dtype=np.int32)
        handle = C.c_void_p()
        nv.check(self.dev._lib.scs_graph_contract(self.dev._ctx, self._h, nv.iptr(gs), len(gs) - 1,
                                                  C.byref(handle)))
        self._h = C.c_void_p()
        return DeviceGraph(self.dev, handle, self.build_stats)

    def download(self) -> np.ndarray:
        n, rb, re_ = self.shape
        out = np.empty((re_ - rb, n))
        nv.check(self.dev._lib.scs_graph_download(self.dev._ctx, self._h, nv.dptr(out)))
        return out

    def download_rows(self, first: int, count: int) -> np.ndarray:
        out = np.empty((count, self.shape[0]))
        nv.check(self.dev._lib.scs_graph_download_rows(self.dev._ctx, self._h, first, count,
                                                       nv.dptr(out)))
        return out

    def degrees(self) -> np.ndarray:
        n, rb, re_ = self.shape
        out = np.empty(re_ - rb)
        nv.check(self.dev._lib.scs_graph_degrees(self.dev._ctx, self._h, nv.dptr(out)))
        return out

    def apply(self, x: np.ndarray) -> np.ndarray:
        """One SYMM launch: rows of S @ x for this rank's row block (x: V x b)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        n, rb, re_ = self.shape
        y = np.empty((re_ - rb, x.shape[1]))
        nv.check(self.dev._lib.scs_debug_apply(self.dev._ctx, self._h, nv.dptr(x), x.shape[1],
                                               nv.dptr(y)))
        return y

    def fiedler(self, x_init: np.ndarray | None = None, tol: float = DEFAULT_TOL,
                max_iter: int = DEFAULT_MAX_ITER, block: int = 0):
        """V x 2 spectral embedding + solver report (reference: scs.py:252).

        Raises ``_native.ConvergenceError`` (carrying the embedding and the report it
        reached) when the residual target was not met.
        """
        n = self.shape[0]
        maps = np.empty((n, 2))
        stats = nv.Stats()
        x0 = None
        if x_init is not None:
            x_init = np.ascontiguousarray(x_init, dtype=np.float64)
            if x_init.shape != (n,):
                msg = f"x_init must have shape ({n},)"
                raise ValueError(msg)
            x0 = nv.dptr(x_init)
        rc = self.dev._lib.scs_fiedler(self.dev._ctx, self._h, x0, tol, max_iter, block,
                                       nv.dptr(maps), C.byref(stats))
        if rc == nv.ENOCONV:
            msg = self.dev._lib.scs_last_error()
            raise nv.ConvergenceError(msg.decode() if msg else "not converged", maps, stats.as_dict())
        nv.check(rc)
        return maps, stats.as_dict()
