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
_state = {"checked": False, "ok": False}


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


def _fast(maps, random_state):
    from sklearn.cluster import _k_means_common as kc
    from sklearn.cluster import _k_means_lloyd as kl
    x = np.array(maps, dtype=np.float64, order="C", copy=True)
    n = x.shape[0]
    # One thread: up to 256 samples are one chunk of the Lloyd kernel anyway, and above that (the
    # fast path ends at 4 096 points x 2 coordinates) an OpenMP team costs far more than it
    # computes -- 6-12 ms per call on a 256-thread host against < 1 ms, for hundreds of
    # recursion nodes.  The chunks are then reduced in index order: deterministic, which the
    # team's order of arrival is not.
    n_threads = 1
    tol = np.mean(np.var(x, axis=0)) * 1e-4
    weight = np.ones(n, dtype=np.float64)
    weight_col = weight.reshape(-1, 1)
    cdf = (weight / weight.sum()).cumsum()
    cdf /= cdf[-1]
    x -= x.mean(axis=0)
    x_sq = np.einsum("ij,ij->i", x, x)
    best_inertia, best_labels = None, None
    seen = {}  # (first, second) seed points -> (labels, inertia): the same start, the same run
    for _ in range(10):
        centres, seeds = _seed_two_centres(x, x_sq, weight, weight_col, cdf, random_state)
        if seeds in seen:
            lab, inertia = seen[seeds]
            if best_inertia is None or (inertia < best_inertia and not kc._is_same_clustering(lab, best_labels, 2)):
                best_labels, best_inertia = lab, inertia
            continue
        # ---- _kmeans_single_lloyd
        centres_new = np.zeros_like(centres)
        lab = np.full(n, -1, dtype=np.int32)
        lab_old = lab.copy()
        in_clusters = np.zeros(2, dtype=np.float64)
        shift = np.zeros(2, dtype=np.float64)
        strict = False
        for _it in range(300):
            kl.lloyd_iter_chunked_dense(x, weight, centres, centres_new, in_clusters, lab, shift, n_threads)
            centres, centres_new = centres_new, centres
            if np.array_equal(lab, lab_old):
                strict = True
                break
            if (shift**2).sum() <= tol:
                break
            lab_old[:] = lab
        if not strict:
            kl.lloyd_iter_chunked_dense(x, weight, centres, centres, in_clusters, lab, shift, n_threads,
                                        update_centers=False)
        inertia = kc._inertia_dense(x, weight, centres, lab, n_threads)
        seen[seeds] = (lab, inertia)
        if best_inertia is None or (inertia < best_inertia and not kc._is_same_clustering(lab, best_labels, 2)):
            best_labels, best_inertia = lab, inertia
    if len(set(best_labels)) < 2:
        import warnings

        from sklearn.exceptions import ConvergenceWarning

        warnings.warn("Number of distinct clusters (1) found smaller than n_clusters (2). Possibly due to "
                      "duplicate points in X.", ConvergenceWarning, stacklevel=3)
    return best_labels


def _self_test() -> bool:
    """The fast path against the public function: labels and generator state, bit for bit."""
    try:
        import warnings

        import sklearn

        if sklearn.__version__ not in _KNOWN:
            return False
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
