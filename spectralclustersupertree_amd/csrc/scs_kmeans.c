/* One run of scikit-learn's Lloyd k-means for TWO clusters of points with TWO coordinates and
 * unit sample weights -- the label assignment of every node of the recursion (reference:
 * src/sc_supertree/scs.py:235-252 -> sklearn/cluster/_spectral.py:759-766 -> k_means(maps, 2,
 * n_init=10)).  Part of libscs_host.so; host code only.
 *
 * kmeans2.py used to drive scikit-learn's compiled iteration (lloyd_iter_chunked_dense) from
 * Python: ten starts x a handful of iterations x ~15 us of call overhead (memoryview set-up, a
 * Python-level row_norms call inside the Cython function) around nanoseconds of arithmetic, for
 * each of the ~60 000 nodes of a 100 000-taxon recursion.  This file restates that iteration --
 * sklearn 1.7.2: _kmeans_single_lloyd (sklearn/cluster/_kmeans.py), lloyd_iter_chunked_dense /
 * _update_chunk_dense (_k_means_lloyd.pyx, one thread), _average_centers, _center_shift,
 * _euclidean_dense_dense, _inertia_dense (_k_means_common.pyx) -- operation by operation:
 *   * the one BLAS call of the iteration, dgemm('t','n', 2, rows, 2, -2, centres, 2, X, 2, 1,
 *     D, 2) on chunks of 256 samples, is made through the SAME function pointer scikit-learn's
 *     Cython code calls (scipy.linalg.cython_blas's exported dgemm; kmeans2.py hands it in), with
 *     the same shapes, so its rounding is not restated but shared;
 *   * everything else is scalar double arithmetic in the order of the Cython source (this file
 *     is compiled with -ffp-contract=off; scikit-learn's wheels target baseline x86-64, no FMA).
 * A start that leaves a cluster empty (scikit-learn then relocates a centre with numpy
 * operations) is not handled here: the function returns 1 and kmeans2.py runs that start on
 * its Python path.  kmeans2.py's self-test compares the whole label assignment with the public
 * k_means bit for bit before any of this is used.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void (*scs_dgemm_fn)(char *, char *, int *, int *, int *, double *, double *, int *, double *,
                             int *, double *, double *, int *);

#define SCS_KM_CHUNK 256 /* CHUNK_SIZE of _k_means_common.pxd */

typedef struct {
    scs_dgemm_fn gemm;
    int n;
    const double *x; /* n x 2, centred */
    double *pw;      /* chunk x 2 scratch */
} km_run;

/* labels (and, with `update`, the new centres / weights / shifts) of one iteration; returns 1
 * when a cluster came out empty */
static int km_iterate(const km_run *r, const double *c_old, double *c_new, double *w_in, int32_t *lab,
                      double *shift, int update) {
    const int n = r->n;
    /* row_norms(centers_old, squared=True): einsum("ij,ij->i") of a 2 x 2 array */
    const double csq0 = c_old[0] * c_old[0] + c_old[1] * c_old[1];
    const double csq1 = c_old[2] * c_old[2] + c_old[3] * c_old[3];
    const int chunk = n > SCS_KM_CHUNK ? SCS_KM_CHUNK : n;
    double cn[4] = {0.0, 0.0, 0.0, 0.0}, wn[2] = {0.0, 0.0}; /* the one thread's local buffers */
    for (int start = 0; start < n; start += chunk) {
        int len = n - start < chunk ? n - start : chunk;
        double *pw = r->pw;
        for (int i = 0; i < len; ++i) {
            pw[2 * i] = csq0;
            pw[2 * i + 1] = csq1;
        }
        char ta = 't', tb = 'n';
        int two = 2;
        double alpha = -2.0, beta = 1.0;
        r->gemm(&ta, &tb, &two, &len, &two, &alpha, (double *)c_old, &two,
                (double *)(r->x + 2 * (size_t)start), &two, &beta, pw, &two);
        for (int i = 0; i < len; ++i) {
            const int label = pw[2 * i + 1] < pw[2 * i] ? 1 : 0;
            lab[start + i] = label;
            if (update) {
                const double *xi = r->x + 2 * (size_t)(start + i);
                wn[label] += 1.0;
                cn[label * 2] += xi[0] * 1.0;
                cn[label * 2 + 1] += xi[1] * 1.0;
            }
        }
    }
    if (!update) return 0;
    for (int j = 0; j < 2; ++j) {
        w_in[j] = 0.0;
        w_in[j] += wn[j];
        for (int k = 0; k < 2; ++k) {
            c_new[j * 2 + k] = 0.0;
            c_new[j * 2 + k] += cn[j * 2 + k];
        }
    }
    if (w_in[0] == 0.0 || w_in[1] == 0.0) return 1; /* _relocate_empty_clusters_dense: Python's */
    for (int j = 0; j < 2; ++j) {
        const double a = 1.0 / w_in[j];
        c_new[j * 2] *= a;
        c_new[j * 2 + 1] *= a;
    }
    for (int j = 0; j < 2; ++j) {
        double res = 0.0;
        const double d0 = c_new[j * 2] - c_old[j * 2], d1 = c_new[j * 2 + 1] - c_old[j * 2 + 1];
        res += d0 * d0;
        res += d1 * d1;
        shift[j] = sqrt(res);
    }
    return 0;
}

/* _kmeans_single_lloyd(X, ones, centres_init, max_iter, tol, n_threads=1) -> labels, inertia.
 * `dgemm`: scipy.linalg.cython_blas's dgemm (Fortran calling convention).  Returns 0, 1 (an
 * empty cluster: run this start elsewhere) or -1 (bad arguments / out of memory). */
int scs_host_lloyd2(void *dgemm, int32_t n, const double *x, const double *centres_init, double tol,
                    int32_t max_iter, int32_t *labels, double *inertia_out, int32_t *n_iter_out) {
    if (!dgemm || n < 1 || !x || !centres_init || !labels || !inertia_out) return -1;
    double pw_small[2 * 64];
    const int chunk = n > SCS_KM_CHUNK ? SCS_KM_CHUNK : n;
    double *pw = chunk <= 64 ? pw_small : (double *)malloc(sizeof(double) * 2 * (size_t)chunk);
    int32_t lab_old_small[64];
    int32_t *lab_old = n <= 64 ? lab_old_small : (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int rc = -1;
    if (pw && lab_old) {
        km_run r = {(scs_dgemm_fn)dgemm, n, x, pw};
        double ca[4], cb[4] = {0.0, 0.0, 0.0, 0.0}, w_in[2] = {0.0, 0.0}, shift[2] = {0.0, 0.0};
        memcpy(ca, centres_init, sizeof ca);
        double *centres = ca, *centres_new = cb;
        for (int i = 0; i < n; ++i) labels[i] = lab_old[i] = -1;
        int strict = 0, it = 0;
        rc = 0;
        for (it = 0; it < max_iter; ++it) {
            if (km_iterate(&r, centres, centres_new, w_in, labels, shift, 1)) {
                rc = 1;
                break;
            }
            double *t = centres;
            centres = centres_new;
            centres_new = t;
            if (memcmp(labels, lab_old, sizeof(int32_t) * (size_t)n) == 0) {
                strict = 1;
                break;
            }
            /* (center_shift ** 2).sum() <= tol */
            if (shift[0] * shift[0] + shift[1] * shift[1] <= tol) break;
            memcpy(lab_old, labels, sizeof(int32_t) * (size_t)n);
        }
        if (rc == 0) {
            if (!strict) km_iterate(&r, centres, centres, w_in, labels, shift, 0);
            /* _inertia_dense, one thread */
            double inertia = 0.0;
            for (int i = 0; i < n; ++i) {
                const double *c = centres + 2 * labels[i], *xi = x + 2 * (size_t)i;
                double res = 0.0;
                const double d0 = xi[0] - c[0], d1 = xi[1] - c[1];
                res += d0 * d0;
                res += d1 * d1;
                inertia += res * 1.0;
            }
            *inertia_out = inertia;
            if (n_iter_out) *n_iter_out = it < max_iter ? it + 1 : max_iter;
        }
    }
    if (pw != pw_small) free(pw);
    if (lab_old != lab_old_small) free(lab_old);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * The whole label assignment of one node: ten k-means++ seedings (sklearn/cluster/_kmeans.py,
 * _kmeans_plusplus with two clusters: 2 + int(log 2) = 2 local trials), the Lloyd run of every
 * distinct start and the choice among them (KMeans.fit) -- kmeans2.py's `_Seeder` and the loop
 * of `_fast`, which remain the definition (and the fallback); this is their restatement for
 * embeddings of at most a few hundred points, where ten seedings cost ~300 numpy calls.
 *
 * The four matrix products of a seeding are numpy `@` calls in scikit-learn.  They are made here
 * through the CBLAS entry points of the OpenBLAS that numpy itself is linked to (kmeans2.py finds
 * them in the library numpy loaded: cblas_dgemv / cblas_ddot / cblas_dgemm, 64-bit integer
 * interface), in the forms that reproduce numpy's results bit for bit:
 *     points (1 x 2) @ x.T          cblas_dgemv(RowMajor, NoTrans, n, 2, 1, x, 2, point, 1, 0, y, 1)
 *     closest (1 x n) @ weight      cblas_ddot(n, closest, 1, ones, 1)
 *     points (2 x 2) @ x.T          cblas_dgemm(RowMajor, NoTrans, Trans, 2, n, 2, 1, pts, 2, x, 2, 0, out, n)
 *     to_cand (2 x n) @ weight_col  cblas_dgemv(RowMajor, NoTrans, 2, n, 1, to_cand, n, ones, 1, 0, y, 1)
 * kmeans2.py's self-test compares the seeds AND the potentials (`dbg`) with the numpy path bitwise
 * on hundreds of inputs before this is used; everything else is scalar arithmetic in numpy's
 * order (elementwise passes, a sequential running sum, binary searches).
 * The thirty uniform draws come from the caller (one RandomState.random_sample call).
 */
typedef void (*cblas_dgemv_fn)(int, int, int64_t, int64_t, double, const double *, int64_t, const double *,
                               int64_t, double, double *, int64_t);
typedef double (*cblas_ddot_fn)(int64_t, const double *, int64_t, const double *, int64_t);
typedef void (*cblas_dgemm_fn)(int, int, int, int64_t, int64_t, int64_t, double, const double *, int64_t,
                               const double *, int64_t, double, double *, int64_t);

typedef struct {
    void *dgemm_f;      /* scipy.linalg.cython_blas dgemm (Fortran interface): the Lloyd iteration's */
    void *cblas_dgemv;  /* numpy's OpenBLAS, ILP64 */
    void *cblas_ddot;
    void *cblas_dgemm;
} scs_km_blas;

enum { KM_ROW_MAJOR = 101, KM_NO_TRANS = 111, KM_TRANS = 112 };

/* d[i] = max(((-2 r[i]) + pp) + x_sq[i], 0): _euclidean_distances(..., squared=True) after the product */
static void km_finish_distances(int n, double *d, double pp, const double *x_sq) {
    for (int i = 0; i < n; ++i) {
        double v = -2.0 * d[i];
        v += pp;
        v += x_sq[i];
        d[i] = v >= 0.0 ? v : 0.0;
    }
}

static int km_same_clustering(const int32_t *a, const int32_t *b, int n) {
    int32_t map[2] = {-1, -1};
    for (int i = 0; i < n; ++i) {
        if (map[a[i]] == -1) map[a[i]] = b[i];
        else if (map[a[i]] != b[i]) return 0;
    }
    return 1;
}

/* Returns 0 (labels_out filled), 1 (a start emptied a cluster: run the call on the Python path
 * with the same draws) or -1.  seeds_out: 2 x starts sample indices; dbg (nullable): per start
 * pot, cand_pot[0], cand_pot[1]. */
int scs_host_kmeans2(const scs_km_blas *blas, int32_t n, const double *x, const double *x_sq,
                     const double *cdf, const double *draws, int32_t starts, double tol, int32_t max_iter,
                     int32_t *labels_out, int32_t *seeds_out, double *dbg) {
    if (!blas || !blas->dgemm_f || !blas->cblas_dgemv || !blas->cblas_ddot || !blas->cblas_dgemm || n < 2 ||
        starts < 1 || starts > 64 || !x || !x_sq || !cdf || !draws || !labels_out)
        return -1;
    const cblas_dgemv_fn gemv = (cblas_dgemv_fn)blas->cblas_dgemv;
    const cblas_ddot_fn dot = (cblas_ddot_fn)blas->cblas_ddot;
    const cblas_dgemm_fn gemm = (cblas_dgemm_fn)blas->cblas_dgemm;
    /* one block: ones[n], closest[n], running[n], to_cand[2n], labels of up to `starts` runs */
    const size_t nd = (size_t)n;
    double *buf = (double *)malloc(sizeof(double) * 5 * nd + sizeof(int32_t) * nd * (size_t)starts);
    if (!buf) return -1;
    double *ones = buf, *closest = buf + nd, *running = buf + 2 * nd, *to_cand = buf + 3 * nd;
    int32_t *run_labels = (int32_t *)(buf + 5 * nd);
    for (int i = 0; i < n; ++i) ones[i] = 1.0;
    int32_t run_first[64], run_second[64];
    double run_inertia[64];
    int n_runs = 0, best = -1, rc = 0, cached_first = -1;
    double pot = 0.0;
    for (int s = 0; s < starts && rc == 0; ++s) {
        const double u0 = draws[3 * s], u1 = draws[3 * s + 1], u2 = draws[3 * s + 2];
        /* first = cdf.searchsorted(u0, side="right"): the number of entries <= u0 */
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u0) lo = mid + 1;
            else hi = mid;
        }
        const int first = lo < n ? lo : n - 1;
        if (first != cached_first) {  /* the distances of a first centre seen in the previous start are kept */
            const double *c = x + 2 * (size_t)first;
            gemv(KM_ROW_MAJOR, KM_NO_TRANS, n, 2, 1.0, x, 2, c, 1, 0.0, closest, 1);
            km_finish_distances(n, closest, c[0] * c[0] + c[1] * c[1], x_sq);
            pot = dot(n, closest, 1, ones, 1);
            double acc = closest[0] * 1.0;
            running[0] = acc;
            for (int i = 1; i < n; ++i) {
                acc += closest[i] * 1.0;
                running[i] = acc;
            }
            cached_first = first;
        }
        int cand[2];
        const double vals[2] = {u1 * pot, u2 * pot};
        for (int k = 0; k < 2; ++k) {  /* searchsorted(running, v), side="left": entries < v */
            int a = 0, b = n;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (running[mid] < vals[k]) a = mid + 1;
                else b = mid;
            }
            cand[k] = a < n - 1 ? a : n - 1;
        }
        double pts[4] = {x[2 * (size_t)cand[0]], x[2 * (size_t)cand[0] + 1], x[2 * (size_t)cand[1]],
                         x[2 * (size_t)cand[1] + 1]};
        gemm(KM_ROW_MAJOR, KM_NO_TRANS, KM_TRANS, 2, n, 2, 1.0, pts, 2, x, 2, 0.0, to_cand, n);
        for (int k = 0; k < 2; ++k) {
            double *row = to_cand + (size_t)k * nd;
            km_finish_distances(n, row, pts[2 * k] * pts[2 * k] + pts[2 * k + 1] * pts[2 * k + 1], x_sq);
            for (int i = 0; i < n; ++i) row[i] = closest[i] <= row[i] ? closest[i] : row[i];  /* np.minimum */
        }
        double cand_pot[2] = {0.0, 0.0};
        gemv(KM_ROW_MAJOR, KM_NO_TRANS, 2, n, 1.0, to_cand, n, ones, 1, 0.0, cand_pot, 1);
        const int second = cand[cand_pot[1] < cand_pot[0] ? 1 : 0];  /* argmin: the first of equals */
        if (seeds_out) {
            seeds_out[2 * s] = first;
            seeds_out[2 * s + 1] = second;
        }
        if (dbg) {
            dbg[3 * s] = pot;
            dbg[3 * s + 1] = cand_pot[0];
            dbg[3 * s + 2] = cand_pot[1];
        }
        /* the same two seed points: the same run */
        int run = -1;
        for (int r = 0; r < n_runs; ++r)
            if (run_first[r] == first && run_second[r] == second) run = r;
        if (run < 0) {
            const double centres[4] = {x[2 * (size_t)first], x[2 * (size_t)first + 1], x[2 * (size_t)second],
                                       x[2 * (size_t)second + 1]};
            run = n_runs;
            const int lrc = scs_host_lloyd2(blas->dgemm_f, n, x, centres, tol, max_iter,
                                            run_labels + (size_t)run * nd, &run_inertia[run], NULL);
            if (lrc != 0) {
                rc = lrc;
                break;
            }
            run_first[run] = first;
            run_second[run] = second;
            ++n_runs;
        }
        if (best < 0 || (run_inertia[run] < run_inertia[best] &&
                         !km_same_clustering(run_labels + (size_t)run * nd, run_labels + (size_t)best * nd, n)))
            best = run;
    }
    if (rc == 0) memcpy(labels_out, run_labels + (size_t)best * nd, sizeof(int32_t) * nd);
    free(buf);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * PROVISIONAL labels of the K nodes of one level of the recursion (round 6, levels.py): the same ten
 * seedings and Lloyd runs on each node's embedding, from draws of the engine's private stream.  These
 * labels only pick the partition the device goes on with; the walk proper assigns every node's labels
 * again with the caller's RandomState through kmeans2.labels and compares (reference order of draws:
 * scs.py:139-171) -- so nothing here has to reproduce numpy's rounding: the centring, the tolerance and the
 * cumulative distribution are computed in plain C.  A node whose seeding empties a cluster is split at the
 * mean of its Fiedler column.
 *   vptr[K + 1]   first vertex of node k in maps / labels_out      maps [vptr[K]][2]      draws [K][30]
 * Returns 0 or -1. */
int scs_host_kmeans2_provisional(const scs_km_blas *blas, int32_t n_nodes, const int64_t *vptr, const double *maps,
                                 const double *draws, int32_t *labels_out) {
    if (!blas || n_nodes < 0 || !vptr || !maps || !draws || !labels_out) return -1;
    int64_t cap = 0;
    for (int32_t k = 0; k < n_nodes; ++k)
        if (vptr[k + 1] - vptr[k] > cap) cap = vptr[k + 1] - vptr[k];
    if (cap == 0) return 0;
    double *buf = (double *)malloc(sizeof(double) * 4 * (size_t)cap);
    if (!buf) return -1;
    double *x = buf, *x_sq = buf + 2 * cap, *cdf = buf + 3 * cap;
    int rc = 0;
    for (int32_t k = 0; k < n_nodes && rc == 0; ++k) {
        const int64_t v0 = vptr[k];
        const int32_t n = (int32_t)(vptr[k + 1] - v0);
        if (n <= 0) continue;
        int32_t *lab = labels_out + v0;
        if (n == 1) {
            lab[0] = 0;
            continue;
        }
        const double *m = maps + 2 * v0;
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < n; ++i) {
            s0 += m[2 * i];
            s1 += m[2 * i + 1];
        }
        const double mean0 = s0 / n, mean1 = s1 / n;
        double q0 = 0.0, q1 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double a = m[2 * i] - mean0, b = m[2 * i + 1] - mean1;
            x[2 * i] = a;
            x[2 * i + 1] = b;
            x_sq[i] = a * a + b * b;
            q0 += a * a;
            q1 += b * b;
            cdf[i] = (double)(i + 1) / n;
        }
        cdf[n - 1] = 1.0;
        const double tol = (q0 / n + q1 / n) / 2.0 * 1e-4;
        const int one = scs_host_kmeans2(blas, n, x, x_sq, cdf, draws + 30 * (size_t)k, 10, tol, 300, lab, NULL, NULL);
        if (one < 0) rc = -1;
        if (one > 0)
            for (int i = 0; i < n; ++i) lab[i] = x[2 * i + 1] > 0.0 ? 1 : 0;
    }
    free(buf);
    return rc;
}
