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

import numpy as np

_KNOWN = ("1.7.2",)
_MAX_SAMPLES = 4096  # larger inputs: the call overhead does not matter
_state = {"checked": False, "ok": False, "native": None}


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


def _seed_two_centres(x, x_sq, weight, weight_col, cdf, random_state):
    """``_kmeans_plusplus`` for two clusters (2 + int(log 2) = 2 local trials).  The first centre
    is ``random_state.choice(n, p=weight / weight.sum())``, spelled out (numpy's legacy
    ``RandomState.choice`` with probabilities: one uniform draw looked up in the normalised
    cumulative sum, ``cdf``, which is the same for all ten initialisations)."""
    centres = np.empty((2, x.shape[1]), dtype=x.dtype)
    first = cdf.searchsorted(random_state.random_sample(), side="right")
    centres[0] = x[first]
    closest = _sq_dist_to(centres[0, np.newaxis], x, x_sq)
    pot = closest @ weight
    rand_vals = random_state.uniform(size=2) * pot
    cand = np.searchsorted(np.cumsum(weight * closest, dtype=np.float64), rand_vals)
    np.clip(cand, None, closest.size - 1, out=cand)
    to_cand = _sq_dist_to(x[cand], x, x_sq)
    np.minimum(closest, to_cand, out=to_cand)
    cand_pot = to_cand @ weight_col
    best = np.argmin(cand_pot)
    second = cand[best]
    centres[1] = x[second]
    return centres, (int(first), int(second))


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


def _fast(maps, random_state, use_native=True):
    from sklearn.cluster import _k_means_common as kc
    from sklearn.cluster import _k_means_lloyd as kl
    x = np.array(maps, dtype=np.float64, order="C", copy=True)
    n = x.shape[0]
    # (one thread in the Lloyd kernel: up to 256 samples are one chunk anyway, and above that -- the
    # fast path ends at 4 096 points x 2 coordinates -- an OpenMP team costs far more than it
    # computes: 6-12 ms per call on a 256-thread host against < 1 ms; the chunks are then reduced
    # in index order, which the team's order of arrival is not)
    tol = np.mean(np.var(x, axis=0)) * 1e-4
    weight = np.ones(n, dtype=np.float64)
    weight_col = weight.reshape(-1, 1)
    cdf = (weight / weight.sum()).cumsum()
    cdf /= cdf[-1]
    x -= x.mean(axis=0)
    x_sq = np.einsum("ij,ij->i", x, x)
    lloyd = _lloyd_cython(x, weight, tol, kl, kc)
    native = _native() if use_native else None
    if native is not None:
        lloyd = _lloyd_native(x, float(tol), lloyd, *native)
    best_inertia, best_labels = None, None
    seen = {}  # (first, second) seed points -> (labels, inertia): the same start, the same run
    for _ in range(10):
        centres, seeds = _seed_two_centres(x, x_sq, weight, weight_col, cdf, random_state)
        if seeds in seen:
            lab, inertia = seen[seeds]
            if best_inertia is None or (inertia < best_inertia and not kc._is_same_clustering(lab, best_labels, 2)):
                best_labels, best_inertia = lab, inertia
            continue
        lab, inertia = lloyd(centres)
        seen[seeds] = (lab, inertia)
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


def _self_test() -> bool:
    """The fast path against the public function: labels and generator state, bit for bit."""
    try:
        import warnings

        import sklearn

        if sklearn.__version__ not in _KNOWN:
            return False
        if _native() is not None and not _native_agrees():
            _state["native"] = False
        probe = np.random.RandomState(12345)
        cases = [probe.standard_normal((n, 2)) * [1.0, 10.0 ** probe.randint(-3, 1)] for n in (3, 4, 5, 7, 12, 33, 100, 300)]
        cases.append(np.array([[0.0, 1.0], [0.0, 1.0], [0.0, -1.0], [0.0, -1.0]]))  # duplicates
        cases.append(np.column_stack([np.full(9, 0.3), np.r_[np.zeros(4), np.ones(5)]]))
        with warnings.catch_warnings():
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


def fast_path_active() -> bool:
    if os.environ.get("SCS_KMEANS", "") == "sklearn":
        return False
    if not _state["checked"]:
        _state["ok"] = _self_test()
        _state["checked"] = True
    return _state["ok"]


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
