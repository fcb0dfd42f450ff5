"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/pcg_oracle.c plus
the numpy pieces that restate the arithmetic around the eigen-solve.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_LIB = Path(__file__).resolve().parent / "libscs_oracle.so"
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not _LIB.exists():
            msg = f"{_LIB} missing: run `make -C oracle` (or __graft_entry__.build())"
            raise ImportError(msg)
        lib = C.CDLL(str(_LIB))
        ip, dp, lp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
        lib.scs_oracle_pcg_dense.restype = C.c_int64
        lib.scs_oracle_pcg_dense.argtypes = [C.c_int32, C.c_int32, C.c_int32, lp, ip, ip, dp, dp, dp]
        lib.scs_oracle_pcg_dense_mt.restype = C.c_int
        lib.scs_oracle_pcg_dense_mt.argtypes = [C.c_int32, C.c_int32, C.c_int32, lp, ip, ip, dp, dp, dp, C.c_int32]
        lib.scs_oracle_pcg_rows.restype = None
        lib.scs_oracle_pcg_rows.argtypes = [C.c_int32, C.c_int32, lp, ip, ip, dp, dp, C.c_int32, ip, ip, dp]
        lib.scs_oracle_contract.restype = None
        lib.scs_oracle_contract.argtypes = [C.c_int32, dp, C.c_int32, ip, dp]
        _lib = lib
    return _lib


def pcg_dense(tables, t_begin: int = 0, t_end: int | None = None, out: np.ndarray | None = None):
    """Dense W of the tables (reference: scs.py:495-663 + :246-250); returns (W, updates)."""
    lib = _load()
    n = tables.n_taxa
    if t_end is None:
        t_end = tables.n_trees
    w = np.zeros((n, n)) if out is None else out
    ip, dp, lp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    updates = lib.scs_oracle_pcg_dense(
        n, t_begin, t_end, tables.tree_off.ctypes.data_as(lp), tables.leaf_taxon.ctypes.data_as(ip),
        tables.adj_depth.ctypes.data_as(ip), tables.adj_val.ctypes.data_as(dp),
        tables.tree_w.ctypes.data_as(dp), w.ctypes.data_as(dp),
    )
    return w, int(updates)


def pcg_dense_mt(tables, threads: int, t_begin: int = 0, t_end: int | None = None,
                 out: np.ndarray | None = None) -> np.ndarray:
    """Dense W on `threads` host threads (rows partitioned, tree order kept per cell): the same
    bits as ``pcg_dense``."""
    lib = _load()
    n = tables.n_taxa
    if t_end is None:
        t_end = tables.n_trees
    w = np.zeros((n, n)) if out is None else out
    ip, dp, lp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    rc = lib.scs_oracle_pcg_dense_mt(
        n, t_begin, t_end, tables.tree_off.ctypes.data_as(lp), tables.leaf_taxon.ctypes.data_as(ip),
        tables.adj_depth.ctypes.data_as(ip), tables.adj_val.ctypes.data_as(dp),
        tables.tree_w.ctypes.data_as(dp), w.ctypes.data_as(dp), int(threads),
    )
    if rc != 0:
        msg = "scs_oracle_pcg_dense_mt failed"
        raise RuntimeError(msg)
    return w


def pcg_rows(tables, rows) -> np.ndarray:
    """Selected rows of W (len(rows) x n_taxa), for spot checks at full size."""
    lib = _load()
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    out = np.zeros((len(rows), tables.n_taxa))
    ip, dp, lp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    lib.scs_oracle_pcg_rows(
        tables.n_taxa, tables.n_trees, tables.tree_off.ctypes.data_as(lp),
        tables.leaf_taxon.ctypes.data_as(ip), tables.adj_depth.ctypes.data_as(ip),
        tables.adj_val.ctypes.data_as(dp), tables.tree_w.ctypes.data_as(dp), len(rows),
        rows.ctypes.data_as(ip), None, out.ctypes.data_as(dp),
    )
    return out


def contract_dense(w: np.ndarray, group_start: np.ndarray) -> np.ndarray:
    lib = _load()
    w = np.ascontiguousarray(w, dtype=np.float64)
    gs = np.ascontiguousarray(group_start, dtype=np.int32)
    ng = len(gs) - 1
    out = np.empty((ng, ng))
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.scs_oracle_contract(w.shape[0], w.ctypes.data_as(dp), ng, gs.ctypes.data_as(ip),
                            out.ctypes.data_as(dp))
    return out


def normalized_operator(a: np.ndarray):
    """S = D^-1/2 A D^-1/2 and dd, as scipy's normalized Laplacian builds them.

    reference: scipy/sparse/csgraph/_laplacian.py:547-558 (m.sum(axis=0); isolated
    rows get dd = 1; two successive divisions).
    """
    m = np.array(a, dtype=np.float64, copy=True)
    np.fill_diagonal(m, 0)
    w = m.sum(axis=0)
    isolated = w == 0
    dd = np.where(isolated, 1, np.sqrt(w))
    m /= dd
    m /= dd[:, np.newaxis]
    return m, dd


def sign_flip_columns(maps: np.ndarray) -> np.ndarray:
    """sklearn/utils/extmath.py:1206-1209 applied to the columns of a V x k array."""
    out = maps.copy()
    for c in range(out.shape[1]):
        i = np.argmax(np.abs(out[:, c]))
        if out[i, c] < 0:
            out[:, c] = -out[:, c]
    return out
