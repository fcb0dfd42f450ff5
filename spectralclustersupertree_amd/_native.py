"""ctypes binding of libscs_hip.so (include/scs_hip.h).  No torch, no fallback.

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C
spectralclustersupertree_amd/csrc``).  If it is missing, or no HIP device is
usable, every entry point raises: there is no CPU path behind this module.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libscs_hip.so"

UNIQUE_ID_BYTES = 128
BUILD_MONOTONE = 1  # SCS_BUILD_MONOTONE
BUILD_SHARED = 2  # SCS_BUILD_SHARED
BUILD_UPPER = 4  # SCS_BUILD_UPPER
BUILD_SCATTER = 8  # SCS_BUILD_SCATTER (comparison variant)


class ScsError(RuntimeError):
    """A libscs_hip call returned a non-zero status."""

    def __init__(self, code: int, message: str) -> None:
        super().__init__(f"libscs_hip error {code}: {message}")
        self.code = code


EINVAL = -1  # SCS_EINVAL
ENOMEM = -3  # SCS_ENOMEM
ENOCONV = -5  # SCS_ENOCONV: scs_fiedler stopped above tol (maps and stats are still filled)
EUNSUP = -6  # SCS_EUNSUP


class ConvergenceError(ScsError):
    """scs_fiedler returned SCS_ENOCONV; ``maps`` and ``stats`` hold what it reached."""

    def __init__(self, message: str, maps, stats: dict) -> None:
        super().__init__(ENOCONV, message)
        self.maps = maps
        self.stats = stats


class Stats(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_int32),
        ("block", C.c_int32),
        ("iterations", C.c_int32),
        ("n_apply", C.c_int32),
        ("converged", C.c_int32),
        ("used_constraint", C.c_int32),
        ("lambda_", C.c_double * 2),
        ("resid", C.c_double * 2),
        ("lambda_next", C.c_double),
        ("apply_ms_total", C.c_double),
        ("apply_ms_min", C.c_double),
        ("solve_ms", C.c_double),
        ("apply_bytes", C.c_double),
        ("allgather_ms_total", C.c_double),
        ("allgather_bytes", C.c_double),
        ("n_allgather", C.c_int32),
        ("n_apply32", C.c_int32),
        ("apply32_ms_total", C.c_double),
        ("apply32_bytes", C.c_double),
        ("lowp_renewals", C.c_int32),
        ("reserved", C.c_int32),
        ("event_pair_ms", C.c_double),
    ]

    def as_dict(self) -> dict:
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name.rstrip("_")] = list(v) if hasattr(v, "__len__") else v
        return out


class ForestInfo(C.Structure):
    _fields_ = [
        ("n_trees", C.c_int32),
        ("monotone", C.c_int32),
        ("n_nodes", C.c_int64),
        ("n_leaves", C.c_int64),
    ]


class BuildStats(C.Structure):
    _fields_ = [
        ("n_taxa", C.c_int32),
        ("n_trees", C.c_int32),
        ("row_begin", C.c_int32),
        ("row_end", C.c_int32),
        ("symmetric", C.c_int32),
        ("n_tiles", C.c_int32),
        ("n_batches", C.c_int32),
        ("spec_batches", C.c_int32),
        ("cell_trees", C.c_double),
        ("prep_ms", C.c_double),
        ("accumulate_ms", C.c_double),
        ("exchange_ms", C.c_double),
        ("total_ms", C.c_double),
        ("bytes_w", C.c_double),
        ("bytes_tables", C.c_double),
        ("exchange_bytes", C.c_double),
        ("tree_parallel_batches", C.c_int32),
        ("spec_trees", C.c_int32),
        ("spec_ms", C.c_double),
        ("listed_batches", C.c_int32),
        ("reserved", C.c_int32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/scs_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_I32 = C.c_int32
# array arguments travel as plain addresses (``ndarray.ctypes.data``): building a typed ctypes pointer
# per argument (``data_as``) costs more than the call itself for the tiny nodes of a deep recursion;
# dptr / iptr / lptr below check the dtype instead
_DP = _IP = _LP = C.c_void_p
ABI_VERSION = 105  # scs_version() of the header these bindings were written against

SIGNATURES = {
    "scs_version": (C.c_int, []),
    "scs_last_error": (C.c_char_p, []),
    "scs_device_count": (C.c_int, []),
    "scs_comm_unique_id": (C.c_int, [_P]),
    "scs_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _PP]),
    "scs_local_group_create": (C.c_int, [C.c_int, _PP]),
    "scs_local_group_destroy": (C.c_int, [_P]),
    "scs_ctx_create_local": (C.c_int, [C.c_int, C.c_int, _P, _PP]),
    "scs_ctx_destroy": (C.c_int, [_P]),
    "scs_ctx_synchronize": (C.c_int, [_P]),
    "scs_ctx_trim": (C.c_int, [_P, C.c_int64]),
    "scs_debug_arena_stats": (C.c_int, [C.c_int, C.POINTER(C.c_int64)]),
    "scs_ctx_reserve": (C.c_int, [_P, C.c_int64]),
    "scs_ctx_comm_info": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "scs_forest_upload": (C.c_int, [_P, _I32, _I32, _LP, _IP, _IP, _DP, _DP, _DP, C.c_int64, _PP]),
    "scs_forest_free": (C.c_int, [_P, _P]),
    "scs_forest_split": (C.c_int, [_P, _P, _IP, _IP, _I32, _IP, _I32, _P, _P]),
    "scs_forest_tables_download": (C.c_int, [_P, _P, _LP, _IP, _IP, _DP, _IP, _DP, _P]),
    "scs_forest_tables_host": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "scs_tables_from_forest": (C.c_int, [_P, _P, _IP, _I32, _PP]),
    "scs_forest_download": (C.c_int, [_P, _P, _I32, _I32, _LP, _IP, _IP, _DP, _DP, _DP]),
    "scs_forest_split_level": (C.c_int, [_P, _P, _IP, _IP, _I32, _I32, _I32, _I32, _IP, _PP, _P, _IP, _LP, _P, _IP, _P]),
    "scs_forest_analyze": (C.c_int, [_P, _P, _IP, _P]),
    "scs_forest_slice": (C.c_int, [_P, _P, _I32, _I32, _PP]),
    "scs_forest_tables_download_range": (C.c_int, [_P, _P, _I32, _I32, _LP, _IP, _IP, _DP, _DP]),
    "scs_tables_from_forest_range": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _IP, _I32, _PP]),
    "scs_small_solve_begin_level": (C.c_int, [_P, _P, _I32, _IP, _IP, _LP, _IP, _IP, _IP, _IP, _IP, _IP, _I32,
                                              C.POINTER(C.c_int32)]),
    "scs_host_alloc": (C.c_int, [C.c_size_t, _PP]),
    "scs_host_free": (C.c_int, [_P]),
    "scs_tables_upload": (C.c_int, [_P, _I32, _I32, _LP, _IP, _IP, _DP, _DP, _PP]),
    "scs_tables_free": (C.c_int, [_P, _P]),
    "scs_pcg_build": (C.c_int, [_P, _P, _I32, _I32, _I32, _PP, C.POINTER(BuildStats)]),
    "scs_graph_contract": (C.c_int, [_P, _P, _IP, _I32, _PP]),
    "scs_graph_matrix_free": (C.c_int, [_P, _P, _I32, _PP]),
    "scs_graph_shape": (C.c_int, [_P, _IP, _IP, _IP]),
    "scs_graph_download": (C.c_int, [_P, _P, _DP]),
    "scs_graph_download_rows": (C.c_int, [_P, _P, _I32, _I32, _DP]),
    "scs_graph_degrees": (C.c_int, [_P, _P, _DP]),
    "scs_graph_free": (C.c_int, [_P, _P]),
    "scs_fiedler": (C.c_int, [_P, _P, _DP, C.c_double, _I32, _I32, _DP, C.POINTER(Stats)]),
    "scs_small_solve": (C.c_int, [_P, _I32, _IP, _IP, _IP, _IP, _IP, _IP, _DP, _DP, _IP, _DP, _DP, _DP]),
    "scs_small_solve_begin": (C.c_int, [_P, _I32, _IP, _IP, _IP, _IP, _IP, _IP, _DP, _DP, _IP, _I32,
                                        C.POINTER(C.c_int32)]),
    "scs_small_solve_begin_forest": (C.c_int, [_P, _P, _IP, _I32, _I32, _IP, _I32, _P]),
    "scs_small_solve_end": (C.c_int, [_P, _I32, _DP, _DP, _DP]),
    "scs_debug_loop_policy": (C.c_int, [C.c_double, _I32, C.c_double, C.c_double, _I32, _I32, _DP, _IP]),
    "scs_debug_jacobi": (C.c_int, [_P, _DP, _I32, _DP, _DP]),
    "scs_debug_gram": (C.c_int, [_P, _DP, _DP, _I32, _I32, _I32, _I32, _DP]),
    "scs_debug_apply": (C.c_int, [_P, _P, _DP, _I32, _DP]),
    "scs_debug_comm_selftest": (C.c_int, [_P, _I32, _DP, _DP]),
    "scs_debug_copy_bandwidth": (C.c_int, [_P, C.c_int64, _I32, _DP]),
}

_lib = None


def load_library() -> C.CDLL:
    """dlopen libscs_hip.so and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        msg = (
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C spectralclustersupertree_amd/csrc`.  There is no CPU fallback."
        )
        raise ImportError(msg)
    lib = C.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    # the ctypes mirrors of the stats structs follow include/scs_hip.h of THIS version: a library
    # built from another header would be handed structs of the wrong size
    have = int(lib.scs_version())
    if have != ABI_VERSION:
        msg = f"{LIB_PATH} reports ABI version {have}, this package binds {ABI_VERSION}: rebuild the library"
        raise ImportError(msg)
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load_library().scs_last_error()
        raise ScsError(rc, msg.decode() if msg else "unknown error")


class _PinnedBlock:
    """Owner of one scs_host_alloc block; freed when the last array viewing it goes away."""

    def __init__(self, nbytes: int) -> None:
        self._lib = load_library()
        self.ptr = C.c_void_p()
        check(self._lib.scs_host_alloc(nbytes, C.byref(self.ptr)))
        self.nbytes = nbytes

    def __del__(self) -> None:
        try:
            if self.ptr:
                self._lib.scs_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """An uninitialised numpy array in page-locked host memory (``scs_host_alloc``): tables
    built in such arrays are uploaded by DMA at the full PCIe rate.  Needs a HIP device."""
    dt = np.dtype(dtype)
    count = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    block = _PinnedBlock(max(count * dt.itemsize, 1))
    buf = (C.c_char * block.nbytes).from_address(block.ptr.value)
    buf._scs_owner = block  # keeps the block alive as long as any view of the buffer lives
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def dptr(a: np.ndarray):
    assert a.dtype == np.float64
    return a.ctypes.data


def iptr(a: np.ndarray):
    assert a.dtype == np.int32
    return a.ctypes.data


def lptr(a: np.ndarray):
    assert a.dtype == np.int64
    return a.ctypes.data
