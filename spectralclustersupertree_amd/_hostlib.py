"""ctypes binding of libscs_host.so (plain-C host helpers: tree arrays, Newick parsing,
contraction groups).  Built by ``__graft_entry__.build()`` / ``make -C csrc``."""

from __future__ import annotations

import ctypes as C
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "libscs_host.so"
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            msg = f"{_LIB_PATH} not found: run __graft_entry__.build()"
            raise ImportError(msg)
        lib = C.CDLL(str(_LIB_PATH))
        # int32 / double / int64 / uint8 arrays, passed as plain addresses (see _native.py)
        ip = dp = lp = bp = C.c_void_p
        lib.scs_host_restrict_sizes.restype = C.c_int
        lib.scs_host_restrict_sizes.argtypes = [C.c_int32, lp, ip, ip, bp, bp, ip]
        lib.scs_host_restrict_fill.restype = C.c_int
        lib.scs_host_restrict_fill.argtypes = [C.c_int32, lp, ip, ip, dp, dp, bp, bp, lp, ip, ip, dp, dp]
        lib.scs_host_split_begin.restype = C.c_int
        lib.scs_host_split_begin.argtypes = [C.c_int32, lp, ip, ip, dp, dp, lp, ip, ip, C.c_int32,
                                             C.POINTER(C.c_void_p), lp, lp]
        lib.scs_host_split_fill.restype = C.c_int
        lib.scs_host_split_fill.argtypes = [C.c_void_p, C.c_int32, lp, ip, lp, ip, ip, dp, dp, bp]
        lib.scs_host_split_end.restype = None
        lib.scs_host_split_end.argtypes = [C.c_void_p]
        lib.scs_host_leaf_counts.restype = C.c_int
        lib.scs_host_leaf_counts.argtypes = [C.c_int32, lp, ip, lp]
        lib.scs_host_flatten.restype = C.c_int
        lib.scs_host_flatten.argtypes = [C.c_int32, lp, ip, ip, dp, dp, C.c_int32, lp, ip, ip, dp, ip, ip]
        lib.scs_host_present.restype = C.c_int
        lib.scs_host_present.argtypes = [C.c_int32, lp, ip, bp]
        lib.scs_host_newick_scan.restype = C.c_int
        lib.scs_host_newick_scan.argtypes = [C.c_char_p, C.c_int64, lp, lp, lp, lp, lp]
        lib.scs_host_newick_parse.restype = C.c_int
        lib.scs_host_newick_parse.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int64, lp, ip, dp, dp,
                                              lp, C.c_char_p, lp, lp]
        lib.scs_host_names_rank.restype = C.c_int
        lib.scs_host_names_rank.argtypes = [C.c_char_p, lp, C.c_int64, ip, lp, lp]
        lib.scs_host_contraction_groups.restype = C.c_int
        lib.scs_host_contraction_groups.argtypes = [C.c_int32, C.c_int32, lp, ip, ip, ip]
        lib.scs_host_components.restype = C.c_int
        lib.scs_host_components.argtypes = [C.c_int32, C.c_int32, lp, ip, ip, ip]
        lib.scs_host_lloyd2.restype = C.c_int
        lib.scs_host_lloyd2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        lib.scs_host_malloc_tune.restype = C.c_int
        lib.scs_host_malloc_tune.argtypes = [C.c_int]
        lib.scs_host_kmeans2.restype = C.c_int
        lib.scs_host_kmeans2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.scs_host_kmeans2_provisional.restype = C.c_int
        lib.scs_host_kmeans2_provisional.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]
        _lib = lib
    return _lib


