"""Label assignment of a bipartition: scikit-learn's k-means (2 clusters, k-means++, 10
initialisations, Lloyd) on the V x 2 embedding, as ``SpectralClustering.fit`` runs it
(sklearn/cluster/_spectral.py:759-766 -> ``k_means(maps, 2, random_state=rs, n_init=10)``).

The recursion makes one such call per node and for all but the top few nodes the call IS the
node: ~1.7 ms, nearly all of it argument validation, array-API dispatch and thread-pool
bookkeeping around a few microseconds of arithmetic (a 20 000-taxon input makes 12 000 calls).
``labels()`` therefore drives scikit-learn's own compiled Lloyd iteration
(``lloyd_iter_chunked_dense``, ``_inertia_dense``, ``_is_same_clustering``) and restates only
the Python glue around it -- centring, the k-means++ seeding, the convergence rule, the choice
among the initialisations (``KMeans.fit``, ``_kmeans_plusplus``, ``_kmeans_single_lloyd`` of
sklearn/cluster/_kmeans.py) -- with the same numpy operations in the same order and the same
draws from the ``RandomState``: same labels, same stream position afterwards.

Round 3: the Lloyd iteration itself -- three quarters of a call for the tiny embeddings of the
deep recursion, all of it call overhead of the Cython entry point -- runs in ``libscs_host.so``
(``csrc/scs_kmeans.c``: the iteration of ``_k_means_lloyd.pyx`` restated for two clusters of
two-coordinate points, its one BLAS call made through the very ``dgemm`` pointer scikit-learn's
Cython code uses).  The self-test holds it against the Cython kernel bit for bit (labels, inertia,
iteration count); without the library, the pointer or a clean self-test the Cython kernel is
driven from Python as before.  ``SCS_KMEANS=cython`` forces that.

Because this leans on private modules of scikit-learn, it is used only when
  * the installed version is one the restatement was written against (``_KNOWN``), and
  * a self-test at first use reproduces the public ``k_means`` bit for bit (labels and the
    generator's state) on a set of probe inputs;
otherwise, and for inputs outside the fast path's range, the public function is called.
``SCS_KMEANS=sklearn`` forces the public function.
"""

from __future__ import annotations

import os
import threading

import numpy as np

_KNOWN = ("1.7.2",)
_MAX_SAMPLES = 4096  # larger inputs: the call overhead does not matter
_NATIVE_WHOLE_MAX = 4096  # the C restatement of the seeding is held against numpy up to this size
_state = {"checked": False, "ok": False, "native": None, "whole": None}
_per_size = {}  # n -> (weight, weight_col, cdf): functions of n alone


def _public(maps, random_state):
    from sklearn.cluster import k_means

    _, lab, _ = k_means(maps, 2, random_state=random_state, n_init=10, verbose=False)
    return np.asarray(lab)


def _sq_dist_to(points, x, x_sq):
    """squared distances from `points` (k x f) to the rows of x, as sklearn's
    ``_euclidean_distances(points, x, Y_norm_squared=x_sq, squared=True)`` forms them"""
    d = -2 * (points @ x.T)
    d += np.einsum("ij,ij->i", points, points)[:, None]
    d += x_sq.reshape(1, -1)
    np.maximum(d, 0, out=d)
    return d


class _Seeder:
    """``_kmeans_plusplus`` for two clusters (2 + int(log 2) = 2 local trials), ten times over.
    The first centre is ``random_state.choice(n, p=weight / weight.sum())``, spelled out (numpy's
    legacy ``RandomState.choice`` with probabilities: one uniform draw looked up in the normalised
    cumulative sum, ``cdf``, which is the same for all ten initialisations).

    An embedding of a handful of points sees the same first centre -- and the same pair of
    candidates for the second -- again and again among the ten starts: everything that depends
    only on the first centre (its distances, their sum and running sum) and only on (first
    centre, candidates) (which candidate wins) is computed once per distinct key, with the very
    numpy calls of the public function.  The thirty uniform draws are taken in one call by the
    caller: ``uniform(size=2)`` is ``0.0 + 1.0 * random_sample(2)``, the same doubles from the same
    stream positions."""

    def __init__(self, x, x_sq, weight, weight_col, cdf, draws):
        self.x, self.x_sq, self.weight, self.weight_col, self.cdf = x, x_sq, weight, weight_col, cdf
        self.draws = draws  # random_state.random_sample(3 * starts)
        self.by_first = {}
        self.by_candidates = {}
        self.at = 0
        self.trace = None  # a list: (first, second, pot, potential of either candidate) per start

    def next(self):
        """(first, second) sample indices of the next start's two centres."""
        x, x_sq = self.x, self.x_sq
        u = self.draws[self.at: self.at + 3]
        self.at += 3
        first = int(self.cdf.searchsorted(u[0], side="right"))
        ent = self.by_first.get(first)
        if ent is None:
            centre = np.empty((2, x.shape[1]), dtype=x.dtype)
            centre[0] = x[first]
            closest = _sq_dist_to(centre[0, np.newaxis], x, x_sq)
            pot = closest @ self.weight
            ent = self.by_first[first] = (closest, pot, np.cumsum(self.weight * closest, dtype=np.float64))
        closest, pot, running = ent
        cand = np.searchsorted(running, u[1:] * pot)
        np.clip(cand, None, closest.size - 1, out=cand)
        key = (first, int(cand[0]), int(cand[1]))
        hit = self.by_candidates.get(key)
        if hit is None:
            to_cand = _sq_dist_to(x[cand], x, x_sq)
            np.minimum(closest, to_cand, out=to_cand)
            cand_pot = to_cand @ self.weight_col
            hit = self.by_candidates[key] = (int(cand[np.argmin(cand_pot)]), float(cand_pot[0, 0]),
                                             float(cand_pot[1, 0]))
        if self.trace is not None:
            self.trace.append((first, hit[0], float(pot[0]), hit[1], hit[2]))
        return first, hit[0]


def _dgemm_pointer():
    """Address of the ``dgemm`` scikit-learn's Cython code calls (``sklearn/utils/_cython_blas.pyx``
    cimports it from ``scipy.linalg.cython_blas``, which exports it in ``__pyx_capi__``)."""
    import ctypes as C

    import scipy.linalg.cython_blas as cb

    cap = cb.__pyx_capi__["dgemm"]
    api = C.pythonapi
    api.PyCapsule_GetName.restype = C.c_char_p
    api.PyCapsule_GetName.argtypes = [C.py_object]
    api.PyCapsule_GetPointer.restype = C.c_void_p
    api.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    name = api.PyCapsule_GetName(cap)
    if b"(char *, char *, int *, int *, int *," not in name:  # not the signature this file calls it with
        return None
    return api.PyCapsule_GetPointer(cap, name)


def _lloyd_cython(x, weight, tol, kl, kc):
    """``_kmeans_single_lloyd`` on scikit-learn's compiled iteration: centres -> (labels, inertia)."""
    n = x.shape[0]
    n_threads = 1

    def run(centres, want_iterations=False):
        centres_new = np.zeros_like(centres)
        lab = np.full(n, -1, dtype=np.int32)
        lab_old = lab.copy()
        in_clusters = np.zeros(2, dtype=np.float64)
        shift = np.zeros(2, dtype=np.float64)
        strict = False
        done = 300
        for it in range(300):
            kl.lloyd_iter_chunked_dense(x, weight, centres, centres_new, in_clusters, lab, shift, n_threads)
            centres, centres_new = centres_new, centres
            if np.array_equal(lab, lab_old):
                strict = True
                done = it + 1
                break
            if (shift**2).sum() <= tol:
                done = it + 1
                break
            lab_old[:] = lab
        if not strict:
            kl.lloyd_iter_chunked_dense(x, weight, centres, centres, in_clusters, lab, shift, n_threads,
                                        update_centers=False)
        inertia = kc._inertia_dense(x, weight, centres, lab, n_threads)
        return (lab, inertia, done) if want_iterations else (lab, inertia)

    return run


def _native():
    """(library, dgemm address) of the C iteration, or None."""
    if _state["native"] is None:
        _state["native"] = False
        if os.environ.get("SCS_KMEANS", "") != "cython":
            try:
                from spectralclustersupertree_amd import _hostlib

                lib, ptr = _hostlib.load(), _dgemm_pointer()
                if ptr:
                    _state["native"] = (lib, ptr)
            except Exception:  # noqa: BLE001 -- no library / no scipy capsule: the Cython kernel it is
                pass
    return _state["native"] or None


def _lloyd_native(x, tol, fallback, lib, ptr):
    """The same on ``scs_host_lloyd2``; a start that empties a cluster goes to `fallback`."""
    import ctypes as C

    n = x.shape[0]
    x_ptr = x.ctypes.data
    c_buf = np.empty((2, 2), dtype=np.float64)
    c_ptr = c_buf.ctypes.data
    inertia = C.c_double()
    iters = C.c_int32()
    inertia_ref, iters_ref = C.byref(inertia), C.byref(iters)
    fn = lib.scs_host_lloyd2

    def run(centres, want_iterations=False):
        c_buf[...] = centres
        lab = np.empty(n, dtype=np.int32)
        rc = fn(ptr, n, x_ptr, c_ptr, tol, 300, lab.ctypes.data, inertia_ref, iters_ref)
        if rc != 0:
            if rc < 0:
                msg = "scs_host_lloyd2 failed"
                raise RuntimeError(msg)
            return fallback(centres, want_iterations)
        return (lab, inertia.value, iters.value) if want_iterations else (lab, inertia.value)

    return run


def _numpy_cblas():
    """Addresses of cblas_dgemv / cblas_ddot / cblas_dgemm (64-bit integer interface) in the
    OpenBLAS numpy itself is linked to -- the routines behind the ``@`` calls of the seeding."""
    import ctypes as C

    from threadpoolctl import threadpool_info

    for info in threadpool_info():
        path = info.get("filepath") or ""
        if info.get("internal_api") != "openblas" or os.path.basename(os.path.dirname(path)) != "numpy.libs":
            continue
        lib = C.CDLL(path)
        found = []
        for name in ("dgemv", "ddot", "dgemm"):
            for sym in (f"scipy_cblas_{name}64_", f"cblas_{name}64_"):
                try:
                    found.append(C.cast(getattr(lib, sym), C.c_void_p).value)
                    break
                except AttributeError:
                    continue
        if len(found) == 3:
            return found
    return None


def _whole():
    """(library, pointer to the struct of BLAS entry points) for ``scs_host_kmeans2``, or None."""
    if _state["whole"] is None:
        _state["whole"] = False
        native = _native()
        if native is not None and os.environ.get("SCS_KMEANS", "") != "seed-numpy":
            try:
                import ctypes as C

                cblas = _numpy_cblas()
                if cblas:
                    table = (C.c_void_p * 4)(native[1], *cblas)
                    _state["whole"] = (native[0], table)
            except Exception:  # noqa: BLE001 -- numpy linked otherwise: the seeding stays numpy's
                pass
    return _state["whole"] or None


def _whole_call(whole, x, x_sq, cdf, draws, tol, want_debug=False):
    """``scs_host_kmeans2``: labels of the best of the ten starts, or None when a start emptied a
    cluster (the caller then runs the numpy / Cython path on the same draws)."""
    lib, table = whole
    n = x.shape[0]
    lab = np.empty(n, dtype=np.int32)
    seeds = dbg = None
    if want_debug:
        seeds, dbg = np.zeros(20, dtype=np.int32), np.zeros(30, dtype=np.float64)
    rc = lib.scs_host_kmeans2(table, n, x.ctypes.data, x_sq.ctypes.data, cdf.ctypes.data, draws.ctypes.data,
                              10, tol, 300, lab.ctypes.data,
                              seeds.ctypes.data if want_debug else None, dbg.ctypes.data if want_debug else None)
    if rc < 0:
        msg = "scs_host_kmeans2 failed"
        raise RuntimeError(msg)
    if want_debug:
        return (lab if rc == 0 else None), seeds, dbg
    return lab if rc == 0 else None


def _centre_and_tolerance(x: np.ndarray):
    """Centres the (n, 2) float64 C-ordered array ``x`` IN PLACE (``x -= x.mean(axis=0)``) and returns
    ``np.mean(np.var(x_before, axis=0)) * 1e-4`` -- the same floating-point operations in the same order as the two
    numpy calls (see ``_fast``), bit for bit."""
    n = x.shape[0]
    mean = np.add.reduce(x, axis=0)
    mean /= n
    x -= mean
    dev = x * x
    var = np.add.reduce(dev, axis=0)
    var /= n
    return (var[0] + var[1]) / 2 * 1e-4


def _fast(maps, random_state, use_native=True):
    from sklearn.cluster import _k_means_common as kc
    from sklearn.cluster import _k_means_lloyd as kl
    x = np.array(maps, dtype=np.float64, order="C", copy=True)
    n = x.shape[0]
    # (one thread in the Lloyd kernel: up to 256 samples are one chunk anyway, and above that -- the
    # fast path ends at 4 096 points x 2 coordinates -- an OpenMP team costs far more than it
    # computes: 6-12 ms per call on a 256-thread host against < 1 ms; the chunks are then reduced
    # in index order, which the team's order of arrival is not)
    # tol = np.mean(np.var(x, axis=0)) * 1e-4 and x -= x.mean(axis=0) (sklearn/cluster/_kmeans.py:_tolerance, :1485),
    # operation for operation as numpy's _var / _mean carry them out (numpy/_core/_methods.py: the column sums over the
    # rows, a true division by the count, the deviations squared in place, their column sums, a true division; the mean
    # of the two variances is their sum halved) -- without the two calls' Python-level bookkeeping, and with the centred
    # array, which both need, made once (``_centre_and_tolerance``; tests/test_kmeans2.py holds it against the calls)
    tol = _centre_and_tolerance(x)
    per_size = _per_size.get(n)
    if per_size is None:
        weight = np.ones(n, dtype=np.float64)
        cdf = (weight / weight.sum()).cumsum()
        cdf /= cdf[-1]
        per_size = (weight, weight.reshape(-1, 1), cdf)
        if n <= 4096:
            _per_size[n] = per_size
    weight, weight_col, cdf = per_size
    x_sq = np.einsum("ij,ij->i", x, x)
    draws = random_state.random_sample(30)
    best_labels = None
    whole = _whole() if use_native and n <= _NATIVE_WHOLE_MAX else None
    if whole is not None:
        best_labels = _whole_call(whole, x, x_sq, cdf, draws, float(tol))
    if best_labels is None:
        lloyd = _lloyd_cython(x, weight, tol, kl, kc)
        native = _native() if use_native else None
        if native is not None:
            lloyd = _lloyd_native(x, float(tol), lloyd, *native)
        best_inertia = None
        seen = {}  # (first, second) seed points -> (labels, inertia): the same start, the same run
        seeder = _Seeder(x, x_sq, weight, weight_col, cdf, draws)
        for _ in range(10):
            seeds = seeder.next()
            run = seen.get(seeds)
            if run is None:
                centres = np.empty((2, 2), dtype=np.float64)
                centres[0] = x[seeds[0]]
                centres[1] = x[seeds[1]]
                run = seen[seeds] = lloyd(centres)
            lab, inertia = run
            if best_inertia is None or (inertia < best_inertia and not kc._is_same_clustering(lab, best_labels, 2)):
                best_labels, best_inertia = lab, inertia
    if len(set(best_labels)) < 2:
        import warnings

        from sklearn.exceptions import ConvergenceWarning

        warnings.warn("Number of distinct clusters (1) found smaller than n_clusters (2). Possibly due to "
                      "duplicate points in X.", ConvergenceWarning, stacklevel=3)
    return best_labels


def _native_agrees() -> bool:
    """The C iteration against scikit-learn's Cython kernel, start by start: labels, inertia and
    iteration count bit for bit -- on tiny, ragged, multi-chunk and duplicate-ridden inputs, and
    from good and bad starting centres.  Also the one formula of the C file that restates a numpy
    call (the centres' squared norms, ``row_norms`` = einsum) on its own."""
    from sklearn.cluster import _k_means_common as kc
    from sklearn.cluster import _k_means_lloyd as kl
    from sklearn.utils.extmath import row_norms

    native = _native()
    if native is None:
        return False
    probe = np.random.RandomState(2718)
    for _ in range(64):
        c = probe.standard_normal((2, 2)) * 10.0 ** probe.randint(-6, 3)
        if not np.array_equal(row_norms(c, squared=True), c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]):
            return False
    sizes = [2, 3, 4, 5, 8, 13, 31, 64, 65, 200, 256, 257, 600, 1025]
    for n in sizes:
        for rep in range(3):
            x = probe.standard_normal((n, 2)) * [1.0, 10.0 ** probe.randint(-4, 1)]
            if rep == 2:
                x[n // 2:] = x[: n - n // 2]  # duplicates
            x = np.ascontiguousarray(x - x.mean(axis=0))
            weight = np.ones(n)
            tol = float(np.mean(np.var(x, axis=0)) * 1e-4)
            ref = _lloyd_cython(x, weight, tol, kl, kc)
            nat = _lloyd_native(x, tol, lambda c, w=False: None, *native)
            for start in range(4):
                centres = x[probe.choice(n, 2, replace=False)].copy() if start < 3 else \
                    probe.standard_normal((2, 2)) * 3.0
                got = nat(centres.copy(), True)
                want = ref(centres.copy(), True)
                if got is None:  # an empty cluster: the C side hands the start back, nothing to compare
                    continue
                if not (np.array_equal(got[0], want[0]) and got[1] == want[1] and got[2] == want[2]):
                    return False
    return True


def _whole_agrees() -> bool:
    """``scs_host_kmeans2`` against the numpy seeding + Cython iteration on the same draws: the
    seed points of all ten starts, the three potentials of every seeding (the results of the BLAS
    calls) and the chosen labels, bit for bit, over the sizes the C path is used for."""
    from sklearn.cluster import _k_means_common as kc
    from sklearn.cluster import _k_means_lloyd as kl

    whole = _whole()
    if whole is None:
        return False
    probe = np.random.RandomState(31415)
    # (every size up to 24: the BLAS kernels' tails; then around their blockings and the Lloyd chunk)
    sizes = list(range(2, 25)) + [31, 32, 33, 47, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 511, 513, 1000,
                                  2048, 4096]
    for n in sizes:
        for rep in range(4):
            x = probe.standard_normal((n, 2)) * [1.0, 10.0 ** probe.randint(-6, 2)]
            if rep == 1:
                x[:, 0] = probe.rand()  # the constant column of an embedding
            if rep == 2:
                x = np.round(x, 1)  # duplicates, ties
            if rep == 3 and n >= 4:
                x[n // 2:] = -x[: n - n // 2]  # mirror symmetry: equal potentials
            x = np.ascontiguousarray(x - x.mean(axis=0))
            tol = float(np.mean(np.var(x, axis=0)) * 1e-4)
            weight = np.ones(n)
            cdf = (weight / weight.sum()).cumsum()
            cdf /= cdf[-1]
            x_sq = np.einsum("ij,ij->i", x, x)
            draws = probe.random_sample(30)
            lab, seeds, dbg = _whole_call(whole, x, x_sq, cdf, draws, tol, want_debug=True)
            seeder = _Seeder(x, x_sq, weight, weight.reshape(-1, 1), cdf, draws)
            seeder.trace = []
            lloyd = _lloyd_cython(x, weight, tol, kl, kc)
            best, best_inertia, seen = None, None, {}
            for _ in range(10):
                sd = seeder.next()
                if lab is None and len(seeder.trace) * 2 > np.count_nonzero(seeds) + 2:
                    break
                if sd not in seen:
                    c = np.empty((2, 2))
                    c[0], c[1] = x[sd[0]], x[sd[1]]
                    seen[sd] = lloyd(c)
                run = seen[sd]
                if best is None or (run[1] < best_inertia and not kc._is_same_clustering(run[0], best, 2)):
                    best, best_inertia = run
            if lab is None:  # a start emptied a cluster: the C side stopped there, nothing chosen
                continue
            for k, (first, second, pot, p0, p1) in enumerate(seeder.trace):
                if (seeds[2 * k], seeds[2 * k + 1]) != (first, second):
                    return False
                if not (dbg[3 * k] == pot and dbg[3 * k + 1] == p0 and dbg[3 * k + 2] == p1):
                    return False
            if not np.array_equal(lab, best):
                return False
    return True


def _self_test() -> bool:
    """The fast path against the public function: labels and generator state, bit for bit."""
    try:
        import warnings

        import sklearn

        if sklearn.__version__ not in _KNOWN:
            return False
        if _native() is not None and not _native_agrees():
            _state["native"] = False
            _state["whole"] = False
        if _whole() is not None and not _whole_agrees():
            _state["whole"] = False
        probe = np.random.RandomState(12345)
        cases = [probe.standard_normal((n, 2)) * [1.0, 10.0 ** probe.randint(-3, 1)] for n in (3, 4, 5, 7, 12, 33, 100, 300)]
        cases.append(np.array([[0.0, 1.0], [0.0, 1.0], [0.0, -1.0], [0.0, -1.0]]))  # duplicates
        cases.append(np.column_stack([np.full(9, 0.3), np.r_[np.zeros(4), np.ones(5)]]))
        # (one OpenMP thread for the public function: with equal inertia among the starts its choice
        # hangs on the summation order of a team-wide reduction and is not repeatable -- DESIGN.md
        # section 1; the one-thread run is the one this module reproduces)
        from threadpoolctl import threadpool_limits

        with warnings.catch_warnings(), threadpool_limits(limits=1, user_api="openmp"):
            warnings.simplefilter("ignore")
            for c in cases:
                ra, rb = np.random.RandomState(7), np.random.RandomState(7)
                la, lb = _fast(c, ra), _public(c, rb)
                if not np.array_equal(la, lb):
                    return False
                sa, sb = ra.get_state(), rb.get_state()
                if not (np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]):
                    return False
        return True
    except Exception:  # noqa: BLE001 -- anything unexpected in private modules: use the public function
        return False


def native_lloyd_active() -> bool:
    """True when the Lloyd iteration runs in libscs_host.so (after the self-test)."""
    return fast_path_active() and bool(_state["native"])


def native_seeding_active() -> bool:
    """True when the embeddings of the fast path's range are labelled by one call into
    libscs_host.so (seedings included)."""
    return fast_path_active() and bool(_state["whole"])


_self_test_lock = threading.Lock()


def fast_path_active() -> bool:
    if os.environ.get("SCS_KMEANS", "") == "sklearn":
        return False
    if not _state["checked"]:
        # ONE thread runs the self-test, the others wait here (without the interpreter lock).  Two
        # ranks of an in-process team reaching their first label assignment together deadlocked:
        # one inside threadpoolctl's dl_iterate_phdr callback (loader lock held, waiting for the
        # interpreter), the other importing a scikit-learn extension (interpreter held, dlopen
        # waiting for the loader lock) -- profiles/r04_team_first_use_deadlock.txt.
        with _self_test_lock:
            if not _state["checked"]:
                _state["ok"] = _self_test()
                _state["checked"] = True
    return _state["ok"]


def provisional_labels(maps, vptr, random_state) -> np.ndarray | None:
    """Labels of K embeddings at once (``maps`` concatenated, node k = rows ``vptr[k] : vptr[k + 1]``) from draws
    of ``random_state`` -- ``scs_host_kmeans2_provisional``: the partition the level-synchronous recursion goes on
    with, NOT the labels of record (``levels.Engine.build`` assigns those with ``labels`` and the caller's
    stream).  None when the C path is not available (the caller then labels node by node)."""
    whole = _whole() if fast_path_active() else None
    if whole is None:
        return None
    lib, table = whole
    vptr = np.ascontiguousarray(vptr, dtype=np.int64)
    k = len(vptr) - 1
    maps = np.ascontiguousarray(maps, dtype=np.float64)
    out = np.zeros(int(vptr[-1]), dtype=np.int32)
    if k == 0 or len(out) == 0:
        return out
    if not np.all(np.isfinite(maps)):
        return None
    draws = random_state.random_sample(30 * k)
    rc = lib.scs_host_kmeans2_provisional(table, k, vptr.ctypes.data, maps.ctypes.data, draws.ctypes.data,
                                          out.ctypes.data)
    if rc != 0:
        return None
    return out


def labels(maps, random_state) -> np.ndarray:
    """Cluster labels (0 / 1) of the rows of `maps`, equal to
    ``sklearn.cluster.k_means(maps, 2, random_state=random_state, n_init=10)[1]``; draws from
    `random_state` exactly what that call draws."""
    maps = np.asarray(maps)
    if (maps.ndim == 2 and 2 <= maps.shape[0] <= _MAX_SAMPLES and maps.shape[1] == 2
            and isinstance(random_state, np.random.RandomState) and np.all(np.isfinite(maps))
            and fast_path_active()):
        return np.asarray(_fast(maps, random_state))
    return _public(maps, random_state)
