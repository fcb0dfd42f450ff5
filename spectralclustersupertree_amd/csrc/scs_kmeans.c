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
