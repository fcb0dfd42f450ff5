// Fiedler solve on gfx950: LOBPCG on S = D^-1/2 A D^-1/2 (A = this rank's row
// block of W, zero diagonal), replacing scikit-learn's shift-invert ARPACK path
// (reference call site: src/sc_supertree/scs.py:235-252; arithmetic replaced:
// sklearn/manifold/_spectral_embedding.py:332-376, scipy _laplacian.py:547-558).
//
// The only kernel that touches the N x N matrix is k_symm (HBM-bound: 8 N^2
// bytes per launch, b/4 flop per byte): one wave streams four rows with 16-byte
// loads, multiplies against the scaled block Z = D^-1/2 X and reduces across
// the 64 lanes with wave shuffles.  Everything else works on N x 3b panels.
// The Rayleigh-Ritz Gram product runs on v_mfma_f64_16x16x4_f64; the 3b x 3b
// eigenproblem is solved on the device by a one-workgroup parallel Jacobi, so
// an iteration needs one small device->host read (the residual norms).

#include <algorithm>
#include <chrono>
#include <cmath>

#include "scs_internal.h"
#include "scs_policy.h"
#include "scs_symm.h"
#include "scs_matfree.h"
#include "scs_symm_tri.h"

#include <sched.h>
#include "scs_panel.h"

constexpr int MAXB = 16;      // widest LOBPCG block
constexpr int MAXS = 64;      // largest matrix the Jacobi kernel takes
constexpr int SLD = MAXS + 1; // LDS leading dimension (breaks the power-of-two stride)

typedef double v4d __attribute__((ext_vector_type(4)));

// zt[k][i] = dinv[i] * x[i*ldx + c0 + k]   (k-major, leading dimension ldz; columns >= n stay 0)
__global__ void k_scale_rows(const double *__restrict__ x, int ldx, int c0, int b, int n,
                             const double *__restrict__ dinv, double *__restrict__ zt,
                             int64_t ldz) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * b) return;
    const int k = idx / n, i = idx - k * n;
    zt[(int64_t)k * ldz + i] = dinv[i] * x[(int64_t)i * ldx + c0 + k];
}

// dst[i*ldd + c0 + k] = src[i*b + k]
__global__ void k_store_cols(const double *__restrict__ src, int b, int n, double *__restrict__ dst,
                             int ldd, int c0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * b) return;
    const int i = idx / b, k = idx - i * b;
    dst[(int64_t)i * ldd + c0 + k] = src[idx];
}

// gathered chunks [world][chunk_rows*b] -> y_full rows by row_splits
__global__ void k_unpack(const double *__restrict__ recv, int64_t chunk, int b,
                         const int32_t *__restrict__ splits, int world, int n,
                         double *__restrict__ y) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * b) return;
    const int i = idx / b, k = idx - i * b;
    int r = 0;
    while (r + 1 < world && i >= splits[r + 1]) ++r;
    y[idx] = recv[(int64_t)r * chunk + (int64_t)(i - splits[r]) * b + k];
}

// ---------------------------------------------------------------------------
// Gram products on tall panels: out (ka x kb) = A^T B, A: n x ka (lda), B: n x kb (ldb)
// Deterministic: per-workgroup partials, then a fixed-order reduction.
// ---------------------------------------------------------------------------
constexpr int GRAM_CH = 64;

__global__ __launch_bounds__(256) void k_gram(const double *__restrict__ a, int lda, int ka,
                                               const double *__restrict__ b, int ldb, int kb,
                                               int n, double *__restrict__ partial) {
    __shared__ double sa[GRAM_CH][3 * MAXB + 1];
    __shared__ double sb[GRAM_CH][3 * MAXB + 1];
    const int tid = threadIdx.x;
    const int nout = ka * kb;
    constexpr int MAXO = (3 * MAXB * 3 * MAXB + 255) / 256;  // outputs per thread
    double acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = 0.0;
    for (int base = blockIdx.x * GRAM_CH; base < n; base += gridDim.x * GRAM_CH) {
        const int rows = min(GRAM_CH, n - base);
        for (int e = tid; e < rows * ka; e += 256) {
            const int r = e / ka, c = e - r * ka;
            sa[r][c] = a[(int64_t)(base + r) * lda + c];
        }
        for (int e = tid; e < rows * kb; e += 256) {
            const int r = e / kb, c = e - r * kb;
            sb[r][c] = b[(int64_t)(base + r) * ldb + c];
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            const int e = tid + o * 256;
            if (e < nout) {
                const int ia = e / kb, jb = e - ia * kb;
                double s = acc[o];
                for (int r = 0; r < rows; ++r) s += sa[r][ia] * sb[r][jb];
                acc[o] = s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        const int e = tid + o * 256;
        if (e < nout) partial[(int64_t)blockIdx.x * nout + e] = acc[o];
    }
}

// fp64 MFMA variant: one wave per workgroup slab, tiles of 16 x 16 outputs.
// A operand lane l: A^T[i = l&15][k = l>>4] = a[row k][col i]; B operand lane l:
// B[k = l>>4][j = l&15]; accumulator reg r of lane l: row (l>>4) + 4r, col l&15.
template <int TA, int TB>
__global__ __launch_bounds__(64) void k_gram_mfma(const double *__restrict__ a, int lda, int ka,
                                                  const double *__restrict__ b, int ldb, int kb,
                                                  int n, double *__restrict__ partial) {
    const int lane = threadIdx.x;
    const int kk = lane >> 4, cc = lane & 15;
    v4d acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int base = blockIdx.x * 4; base < n; base += gridDim.x * 4) {
        const int r = base + kk;
        double fa[TA], fb[TB];
#pragma unroll
        for (int i = 0; i < TA; ++i) {
            const int c = i * 16 + cc;
            fa[i] = (r < n && c < ka) ? a[(int64_t)r * lda + c] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int c = j * 16 + cc;
            fb[j] = (r < n && c < kb) ? b[(int64_t)r * ldb + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    const int nout = ka * kb;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + kk + 4 * r;
                const int col = j * 16 + cc;
                if (row < ka && col < kb)
                    partial[(int64_t)blockIdx.x * nout + row * kb + col] = acc[i][j][r];
            }
}

// one wave per output element: lanes read the partials (fixed assignment), then a
// fixed-order shuffle tree -> deterministic result
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partial,
                                                          int nparts, int nout,
                                                          double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= nout) return;
    double s = 0.0;
    for (int p = lane; p < nparts; p += 64) s += partial[(int64_t)p * nout + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[e] = s;
}

// ---------------------------------------------------------------------------
// panel updates (row-local, safe in place)
// ---------------------------------------------------------------------------
// y[:, 0:kc] = alpha_y * y[:, 0:kc] + sign * a[:, 0:ka] * c   (c: ka x kc, row-major, ldc)
// y and a may alias (in-place y = y*c): every thread of a row reads the row's inputs
// before any thread of that row writes, because a row lives inside one workgroup
// and a barrier separates the two phases.
__global__ __launch_bounds__(256) void k_update(double *y, int ldy, int kc, double alpha_y,
                                                 const double *a, int lda, int ka,
                                                 const double *__restrict__ c, int ldc,
                                                 double sign, int n) {
    __shared__ double sc[3 * MAXB][MAXB];
    for (int e = threadIdx.x; e < ka * kc; e += 256) {
        const int i = e / kc, j = e - i * kc;
        sc[i][j] = c[i * ldc + j];
    }
    __syncthreads();
    // 256 / 16 = 16 rows per workgroup, 16 column slots per row
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int j = threadIdx.x & 15;
    double acc = 0.0, yold = 0.0;
    const bool live = r < n && j < kc;
    if (live) {
        const double *ar = a + (int64_t)r * lda;
        for (int i = 0; i < ka; ++i) acc += ar[i] * sc[i][j];
        if (alpha_y != 0.0) yold = y[(int64_t)r * ldy + j];
    }
    __syncthreads();
    if (live) y[(int64_t)r * ldy + j] = (alpha_y == 0.0 ? 0.0 : alpha_y * yold) + sign * acc;
}

// Rayleigh-Ritz update of a panel q = [X | R | P] (n x 3b, ld 3b): with coefficient
// matrices c, d (nq x b each, nq = 2b or 3b)   X' = q[:, :nq] c,   P' = q[:, :nq] d,
// written in place (a row lives inside one workgroup; a barrier separates reads from writes).
__global__ __launch_bounds__(256) void k_rr_update(double *q, int b, int nq,
                                                    const double *__restrict__ c,
                                                    const double *__restrict__ d, int n) {
    __shared__ double sc[3 * MAXB][MAXB];
    __shared__ double sd[3 * MAXB][MAXB];
    for (int e = threadIdx.x; e < nq * b; e += 256) {
        const int i = e / b, j = e - i * b;
        sc[i][j] = c[e];
        sd[i][j] = d[e];
    }
    __syncthreads();
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int j = threadIdx.x & 15;
    const bool live = r < n && j < b;
    double xa = 0.0, pa = 0.0;
    double *qr = q + (int64_t)(live ? r : 0) * (3 * b);
    if (live) {
        for (int i = 0; i < nq; ++i) {
            const double v = qr[i];
            xa += v * sc[i][j];
            pa += v * sd[i][j];
        }
    }
    __syncthreads();
    if (live) {
        qr[j] = xa;
        qr[b + j] = pa;
    }
}

// R = AX - X diag(theta) into q[:, 2b:3b]; per-workgroup partial squared norms
__global__ __launch_bounds__(256) void k_residual(double *__restrict__ q,
                                                   const double *__restrict__ aq, int b,
                                                   const double *__restrict__ theta, int n,
                                                   double *__restrict__ partial) {
    __shared__ double red[4][MAXB];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double sq[MAXB];
#pragma unroll
    for (int j = 0; j < MAXB; ++j) sq[j] = 0.0;
    if (r < n) {
        double *qr = q + (int64_t)r * (3 * b);
        const double *ar = aq + (int64_t)r * (3 * b);
#pragma unroll
        for (int j = 0; j < MAXB; ++j)
            if (j < b) {
                const double v = ar[j] - qr[j] * theta[j];
                qr[2 * b + j] = v;
                sq[j] = v * v;
            }
    }
#pragma unroll
    for (int j = 0; j < MAXB; ++j) {
        double s = sq[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < b)
        partial[(int64_t)blockIdx.x * b + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] +
                                                         red[2][threadIdx.x] + red[3][threadIdx.x];
}

__device__ __forceinline__ double hash_uniform(unsigned long long i, unsigned long long j) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull + j * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (double)(x >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

// q[:, 0:b] = [x_init | hashed uniform(-1,1)]
__global__ void k_init_block(double *__restrict__ q, int b, int n, const double *__restrict__ x0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * b) return;
    const int i = idx / b, k = idx - i * b;
    double v = hash_uniform((unsigned long long)i, (unsigned long long)k + 1);
    if (k == 0 && x0) v = x0[i];
    q[(int64_t)i * (3 * b) + k] = v;
}

__global__ void k_fill_int(int *p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// u[i] = sqrt(deg[i]) / ||sqrt(deg)||   (trivial eigenvector of S when no row is isolated)
__global__ void k_trivial(const double *__restrict__ deg, double inv_norm, int n,
                          double *__restrict__ u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) u[i] = sqrt(deg[i]) * inv_norm;
}

// The two columns the caller gets, taken from the panel on the device: the trivial eigenvector
// (constant 1 / ||sqrt d|| when it was deflated analytically, else X's first column / sqrt d) and
// the Fiedler column / sqrt d (sklearn/manifold/_spectral_embedding.py:463).  16 V bytes travel
// instead of the 8 V 3b of the whole panel.
__global__ void k_extract_maps(const double *__restrict__ q, int ldq, int n,
                               const double *__restrict__ dinv, double inv_norm, int constrained,
                               double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double di = dinv[i];
    if (constrained) {
        out[2 * i] = inv_norm;  // (sqrt(d_i) / ||sqrt d||) / sqrt(d_i)
        out[2 * i + 1] = q[(int64_t)i * ldq] * di;
    } else {
        out[2 * i] = q[(int64_t)i * ldq] * di;
        out[2 * i + 1] = q[(int64_t)i * ldq + 1] * di;
    }
}

// SCS_BUILD_UPPER jobs: y = dinv (.) (sum over ranks, in rank order, of the gathered partial
// products); parts is world x (n * b)
__global__ void k_sum_parts(const double *__restrict__ parts, int world, int n, int b,
                            const double *__restrict__ dinv, double *__restrict__ y) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * b) return;
    double s = 0.0;
    for (int r = 0; r < world; ++r) s += parts[(int64_t)r * n * b + idx];
    y[idx] = dinv[idx / b] * s;
}

// ---------------------------------------------------------------------------
// one-workgroup parallel Jacobi eigensolver (n <= 64), LDS resident
// ---------------------------------------------------------------------------
struct jacobi_lds {
    alignas(16) double a[MAXS][SLD];
    alignas(16) double e[MAXS][SLD];
    alignas(16) double cs[MAXS / 2][2];
    int pq[MAXS / 2][2];
    double red[256];
    double w[MAXS];
    int perm[MAXS];
    __device__ double *rot_log() { return &e[16][0]; }  // jacobi_eig_waves' log of a sweep's rotations
};

// The same for ONE size M in {4, 8}: 2 KB instead of 80 -- a kernel whose LDS also holds another role's
// tiles (k_symm_tri_tf) runs its small solve on this one (jacobi_eig_waves is written against either)
template <int M>
struct jacobi_small_lds {
    alignas(16) double a[M][M];
    alignas(16) double e[M][M];
    alignas(16) double cs[M / 2][2];
    alignas(16) double rlog[2 * (M - 1) * (M / 2) * 2];
    double w[M];
    int pq[1][2];
    int perm[M];
    __device__ double *rot_log() { return rlog; }
};

// On entry s.a holds the symmetric matrix (n x n).  On exit s.w holds the
// eigenvalues in DESCENDING order and s.perm[k] the column of s.e holding the
// k-th eigenvector.  All 256 threads of the workgroup must call.
//
// Parallel-ordered two-sided Jacobi.  A step rotates m/2 disjoint index pairs at
// once: A <- J^T A J splits into independent 2x2 blocks (row pair x column pair),
// each updated by one thread, so a step costs two barriers.
// ---- fast variant for n <= 24 (the sizes the fused LOBPCG loop solves every iteration)
// A lone wave issues about one instruction per 8 clocks, so a Jacobi step is bound by the
// instruction count on its critical path, not by arithmetic.  This variant therefore
//   * keeps the rotated pairs at FIXED positions (2k, 2k+1) and moves the data instead: every
//     work item -- a 2 x 2 block of A or a column pair of one row of E -- reads from one copy
//     of the matrices and writes its results to precomputed, round-robin-permuted addresses
//     in the other copy, so a step needs no index arithmetic;
//   * computes each pair's rotation once (m / 2 lanes, from the pair's diagonal block, with
//     hardware rsq/rcp seeds refined by Newton steps instead of IEEE division and square
//     root) and hands it to the items through LDS: a second barrier per step, but the long
//     rotation chain is off the path of the threads that own two items.
// At most two items per thread (256 threads).  Interface as jacobi_eig.
// 1 / sqrt(x) to full precision: the hardware seed (2^-23) and ONE third-order step
// y (1 + h / 2 + 3 h^2 / 8), h = 1 - x y^2 -- four dependent operations
__device__ __forceinline__ double rsq_full(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double xy = x * y;
    const double h = fma(-xy, y, 1.0);
    const double p = fma(0.375, h, 0.5);
    return fma(y * h, p, y);
}

// The rotation (c, s) that annihilates a_pq, |angle| <= pi / 4:  tan 2t = a_pq / dl with
// dl = (a_qq - a_pp) / 2, taken through cos 2t = |dl| / hypot(dl, a_pq):
//   c^2 = (1 + cos 2t) / 2,   s = sin 2t / (2 c).
// A Jacobi step of a lone wave is bound by the LENGTH of this dependent chain (about ten clocks
// an operation), so it is written for depth: two reciprocal square roots, no division, 17
// operations deep (the textbook t = sign / (|theta| + sqrt(theta^2 + 1)) form with refined
// rsq / rcp seeds was 31).  Straight-line (selects, no branch).
__device__ __forceinline__ void jacobi_rot(double app, double apq, double aqq, double &c,
                                           double &sn) {
    const double dl = 0.5 * (aqq - app);
    const double x = fma(dl, dl, apq * apq);
    const double y = rsq_full(x);                   // 1 / hypot(dl, apq)
    const double c2 = fma(0.5 * fabs(dl), y, 0.5);  // in [0.5, 1]
    const double z = rsq_full(c2);
    const double hs = (0.5 * apq) * y;
    const double cc = c2 * z;
    double ss = hs * z;
    ss = dl >= 0.0 ? ss : -ss;
    // nothing to rotate, or hypot^2 outside the range the seed handles (never seen: the
    // matrices are O(1) and dead directions carry -1e30): identity
    const bool rot = apq != 0.0 && x >= 1e-290 && x <= 1e290;
    c = rot ? cc : 1.0;
    sn = rot ? ss : 0.0;
}

__device__ void jacobi_eig_fast(jacobi_lds &s, int n) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = n + (n & 1), half = m / 2;
    const int nblk = half * half, ne = m * half;
    double *fa = &s.a[0][0];  // [2][nblk][4]: 2 x 2 blocks, row-major inside a block
    double *fe = &s.e[0][0];  // [2][ne][2]:   E[r][2 kc], E[r][2 kc + 1]
    // position after one round-robin move: 0 stays, the rest cycles 1 -> 3 -> ... -> m-1 ->
    // m-2 -> ... -> 4 -> 2 -> 1
    auto next_pos = [&](int i) {
        if (i == 0 || m == 2) return i;
        if (i & 1) return i == m - 1 ? m - 2 : i + 2;
        return i == 2 ? 1 : i - 2;
    };
    int kind[2], ia[2], ib[2], dst[2][4];
    double v[2][4];
    // blocks take whole waves (a wave that mixes blocks and E items would run both code
    // paths one after the other), E items follow
    const int nblk_pad = (nblk + 63) & ~63;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int e = tid + 256 * w;
        kind[w] = 0;
        ia[w] = ib[w] = 0;
        if (e < nblk) {
            kind[w] = 1;
            const int kr = e / half, kc = e - kr * half;
            ia[w] = kr;
            ib[w] = kc;
#pragma unroll
            for (int al = 0; al < 2; ++al)
#pragma unroll
                for (int be = 0; be < 2; ++be) {
                    const int i = 2 * kr + al, j = 2 * kc + be;
                    v[w][al * 2 + be] = (i < n && j < n) ? s.a[i][j] : 0.0;
                    const int pi = next_pos(i), pj = next_pos(j);
                    dst[w][al * 2 + be] = ((pi >> 1) * half + (pj >> 1)) * 4 + (pi & 1) * 2 + (pj & 1);
                }
        } else if (e >= nblk_pad && e < nblk_pad + ne) {
            kind[w] = 2;
            const int r = (e - nblk_pad) / half, kc = (e - nblk_pad) - r * half;
            ia[w] = r;
            ib[w] = kc;
#pragma unroll
            for (int be = 0; be < 2; ++be) {
                const int j = 2 * kc + be;
                v[w][be] = r == j ? 1.0 : 0.0;
                const int pj = next_pos(j);
                dst[w][be] = (r * half + (pj >> 1)) * 2 + (pj & 1);
            }
        }
    }
    __syncthreads();  // everyone has read its part of the input matrix
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        if (kind[w] == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) fa[(ia[w] * half + ib[w]) * 4 + k] = v[w][k];
        } else if (kind[w] == 2) {
            fe[(ia[w] * half + ib[w]) * 2 + 0] = v[w][0];
            fe[(ia[w] * half + ib[w]) * 2 + 1] = v[w][1];
        }
    }
    __syncthreads();
    int cur = 0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double *ac = fa + cur * nblk * 4;
        // convergence: off-diagonal mass against the diagonal (dead directions carry -1e30 on
        // the diagonal, k_small_rr: not part of the scale)
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int w = 0; w < 2; ++w)
            if (kind[w] == 1) {
                const double *bp = ac + (ia[w] * half + ib[w]) * 4;
                const double b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
                if (ia[w] == ib[w]) {
                    dia += (b0 > -1e29 ? b0 * b0 : 0.0) + (b3 > -1e29 ? b3 * b3 : 0.0);
                    off += b1 * b1 + b2 * b2;
                } else {
                    off += (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_xor(off, o, 64);
            dia += __shfl_xor(dia, o, 64);
        }
        if (lane == 0) {
            s.red[wave] = off;
            s.red[4 + wave] = dia;
        }
        __syncthreads();
        off = s.red[0] + s.red[1] + s.red[2] + s.red[3];
        dia = s.red[4] + s.red[5] + s.red[6] + s.red[7];
        __syncthreads();
        if (off <= 1.25e-32 * (double)(n * n) * dia || off == 0.0) break;

        for (int step = 0; step < m - 1; ++step) {
            const double *a0 = fa + cur * nblk * 4;
            const double *e0 = fe + cur * ne * 2;
            double *a1 = fa + (cur ^ 1) * nblk * 4;
            double *e1 = fe + (cur ^ 1) * ne * 2;
            // the rotation of pair k, once, from its diagonal block
            if (tid < half) {
                const double *dc = a0 + (tid * half + tid) * 4;
                double c, sn;
                jacobi_rot(dc[0], dc[1], dc[3], c, sn);
                s.cs[tid][0] = c;
                s.cs[tid][1] = sn;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                if (kind[w] == 0) continue;
                const double c2 = s.cs[ib[w]][0], s2 = s.cs[ib[w]][1];
                if (kind[w] == 1) {
                    const double c1 = s.cs[ia[w]][0], s1 = s.cs[ia[w]][1];
                    const double *bp = a0 + (ia[w] * half + ib[w]) * 4;
                    const double a00 = bp[0], a01 = bp[1], a10 = bp[2], a11 = bp[3];
                    const double t00 = c2 * a00 - s2 * a01, t01 = s2 * a00 + c2 * a01;
                    const double t10 = c2 * a10 - s2 * a11, t11 = s2 * a10 + c2 * a11;
                    a1[dst[w][0]] = c1 * t00 - s1 * t10;
                    a1[dst[w][1]] = c1 * t01 - s1 * t11;
                    a1[dst[w][2]] = s1 * t00 + c1 * t10;
                    a1[dst[w][3]] = s1 * t01 + c1 * t11;
                } else {
                    const double *ep = e0 + (ia[w] * half + ib[w]) * 2;
                    const double x = ep[0], y = ep[1];
                    e1[dst[w][0]] = c2 * x - s2 * y;
                    e1[dst[w][1]] = s2 * x + c2 * y;
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    // back to the caller's layout: eigenvector of position j in column j of s.e, positions
    // ranked by eigenvalue (descending); an odd n leaves out the padded coordinate, whose
    // column is the only one with a non-zero in row n
    const double *ac = fa + cur * nblk * 4;
    const double *ec = fe + cur * ne * 2;
#pragma unroll
    for (int w = 0; w < 2; ++w)
        if (kind[w] == 2) {
            v[w][0] = ec[(ia[w] * half + ib[w]) * 2 + 0];
            v[w][1] = ec[(ia[w] * half + ib[w]) * 2 + 1];
        }
    double wi = 0.0;
    int rank = -1;
    if (tid < m) {
        auto diag_at = [&](int i) { return ac[((i >> 1) * half + (i >> 1)) * 4 + (i & 1) * 3]; };
        auto is_pad = [&](int i) { return m != n && ec[(n * half + (i >> 1)) * 2 + (i & 1)] != 0.0; };
        if (!is_pad(tid)) {
            wi = diag_at(tid);
            rank = 0;
            for (int j = 0; j < m; ++j) {
                if (is_pad(j)) continue;
                const double wj = diag_at(j);
                if (wj > wi || (wj == wi && j < tid)) ++rank;
            }
        }
    }
    __syncthreads();  // all reads of the flat copies are done: s.e may be overwritten
#pragma unroll
    for (int w = 0; w < 2; ++w)
        if (kind[w] == 2 && ia[w] < n) {
            s.e[ia[w]][2 * ib[w] + 0] = v[w][0];
            s.e[ia[w]][2 * ib[w] + 1] = v[w][1];
        }
    if (rank >= 0) {
        s.w[rank] = wi;
        s.perm[rank] = tid;
    }
    __syncthreads();
}

// (Round 4: a ONE-wave variant of the above for n <= 12 -- two items per lane, no workgroup
// barrier inside the sweeps, bitwise the same results -- was built and measured slower:
// k_small_rr 50.6 us against 43.5, k_small_orth unchanged at 13.1: four waves with one item per
// thread and two cheap barriers beat one wave that serialises two items and both item kinds;
// profiles/r04_jacobi_wave.txt.  What did work is to split the two item KINDS over two waves:
// jacobi_eig_waves below.)
// The same sum when only the first 16 (ROWS = 1) or 32 (ROWS = 2) lanes carry non-zero terms and
// only they need the result: four DPP steps inside a row of 16 lanes (quad swaps, then the two
// mirrors) instead of six trips through the LDS crossbar; every lane of the rows that count
// ends with bitwise the same total.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int ROWS>
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    if (ROWS == 2) v += __shfl_xor(v, 16, 64);
    return v;
}

// ---- wave-specialised variant for n = 4, 8, 12 (the sizes the fused b = 4 loop solves in
// every iteration: the 12 x 12 Rayleigh-Ritz problem and the 4 x 4 SVQB ones)
// The variant above spends 1 200 clocks on a step at n = 12 -- two workgroup barriers, three
// dependent LDS round trips, the rotation chain -- for 36 + 72 work items.  Here wave 0 alone
// iterates on A: its lanes own the (n/2)^2 2 x 2 blocks in registers, every lane computes the
// rotation of its own block (only the diagonal blocks' ones mean anything: they are published,
// 16 bytes each), applies the two it needs and writes its four results to their round-robin
// positions, in place -- a wave's LDS operations execute in order, so the step needs no barrier
// at all.  The rotations of a sweep are also logged; wave 1, one lane per ROW of the eigenvector
// matrix with the row in registers (the round-robin moves are compile-time renamings over the
// unrolled sweep), replays the log one sweep behind wave 0, concurrently: one workgroup barrier
// per sweep.  Same rotations, same arithmetic, same order as the variant above.
typedef double jw_d2 __attribute__((ext_vector_type(2)));
constexpr int jw_next_pos(int i, int m) {
    if (i == 0 || m == 2) return i;
    if (i & 1) return i == m - 1 ? m - 2 : i + 2;
    return i == 2 ? 1 : i - 2;
}

template <int M, typename L>
__device__ void jacobi_eig_waves(L &s) {
    constexpr int H = M / 2, NB = H * H, SPS = M - 1;
    static_assert(M % 2 == 0 && NB <= 64 && M <= 16, "one wave holds the blocks");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    double *fa = &s.a[0][0];                 // [NB][4]: 2 x 2 blocks, row-major inside a block
    double *rlog = s.rot_log();              // [2][SPS][H][2]: (c, s) of a sweep's rotations
    volatile int *flag = &s.pq[0][0];        // [2]: wave 0's verdict before sweep k, at k & 1
    const int kr = lane / H, kc = lane - kr * H;
    const bool blk = wave == 0 && lane < NB;
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
    int d00 = 0, d01 = 0, d10 = 0, d11 = 0;
    if (blk) {
        a00 = s.a[2 * kr][2 * kc];
        a01 = s.a[2 * kr][2 * kc + 1];
        a10 = s.a[2 * kr + 1][2 * kc];
        a11 = s.a[2 * kr + 1][2 * kc + 1];
        const int r0 = jw_next_pos(2 * kr, M), r1 = jw_next_pos(2 * kr + 1, M);
        const int c0 = jw_next_pos(2 * kc, M), c1 = jw_next_pos(2 * kc + 1, M);
        auto at = [&](int pi, int pj) { return ((pi >> 1) * H + (pj >> 1)) * 4 + (pi & 1) * 2 + (pj & 1); };
        d00 = at(r0, c0);
        d01 = at(r0, c1);
        d10 = at(r1, c0);
        d11 = at(r1, c1);
    }
    // row `lane` of E (wave 1), by position
    double e[M];
#pragma unroll
    for (int i = 0; i < M; ++i) e[i] = i == lane ? 1.0 : 0.0;
    __syncthreads();  // the input matrix has been read: its memory now holds the flat blocks
    if (blk) {
        double *bp = fa + lane * 4;
        bp[0] = a00;
        bp[1] = a01;
        bp[2] = a10;
        bp[3] = a11;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        if (wave == 0) {
            // convergence: off-diagonal mass against the diagonal (dead directions carry -1e30 on
            // the diagonal, k_small_rr: not part of the scale)
            double off = 0.0, dia = 0.0;
            if (blk) {
                if (kr == kc) {
                    dia = (a00 > -1e29 ? a00 * a00 : 0.0) + (a11 > -1e29 ? a11 * a11 : 0.0);
                    off = a01 * a01 + a10 * a10;
                } else {
                    off = (a00 * a00 + a01 * a01) + (a10 * a10 + a11 * a11);
                }
            }
            off = row_sum<1>(off);
            dia = row_sum<1>(dia);
            if (NB > 16) {
                off += __shfl_xor(off, 16, 64);
                dia += __shfl_xor(dia, 16, 64);
            }
            if (NB > 32) {
                off += __shfl_xor(off, 32, 64);
                dia += __shfl_xor(dia, 32, 64);
            }
            const bool done = off <= 1.25e-32 * (double)(M * M) * dia || off == 0.0;
            if (lane == 0) flag[sweep & 1] = done ? 1 : 0;
            if (!done) {
                double *lg = rlog + (sweep & 1) * SPS * H * 2;
                for (int step = 0; step < SPS; ++step) {
                    double c, sn;
                    jacobi_rot(a00, a01, a11, c, sn);
                    if (blk && kr == kc) {
                        const jw_d2 v = {c, sn};
                        *(jw_d2 *)&s.cs[kr][0] = v;
                        *(jw_d2 *)&lg[(step * H + kr) * 2] = v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (blk) {
                        const jw_d2 r1 = *(const jw_d2 *)&s.cs[kr][0], r2 = *(const jw_d2 *)&s.cs[kc][0];
                        const double c1 = r1.x, s1 = r1.y;
                        const double c2 = r2.x, s2 = r2.y;
                        const double t00 = c2 * a00 - s2 * a01, t01 = s2 * a00 + c2 * a01;
                        const double t10 = c2 * a10 - s2 * a11, t11 = s2 * a10 + c2 * a11;
                        fa[d00] = c1 * t00 - s1 * t10;
                        fa[d01] = c1 * t01 - s1 * t11;
                        fa[d10] = s1 * t00 + c1 * t10;
                        fa[d11] = s1 * t01 + c1 * t11;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (blk) {
                        const jw_d2 *bp = (const jw_d2 *)(fa + lane * 4);
                        const jw_d2 lo = bp[0], hi = bp[1];
                        a00 = lo.x;
                        a01 = lo.y;
                        a10 = hi.x;
                        a11 = hi.y;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
        }
        __syncthreads();
        if (flag[sweep & 1]) break;
        if (wave == 1 && lane < M) {
            const double *lg = rlog + (sweep & 1) * SPS * H * 2;
#pragma unroll
            for (int step = 0; step < SPS; ++step) {
                double t[M];
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    const jw_d2 r = *(const jw_d2 *)&lg[(step * H + k) * 2];
                    const double c2 = r.x, s2 = r.y;
                    const double x = e[2 * k], y = e[2 * k + 1];
                    t[jw_next_pos(2 * k, M)] = c2 * x - s2 * y;
                    t[jw_next_pos(2 * k + 1, M)] = s2 * x + c2 * y;
                }
#pragma unroll
                for (int i = 0; i < M; ++i) e[i] = t[i];
            }
        }
    }
    // back to the caller's layout: eigenvector of position j in column j of s.e, positions
    // ranked by eigenvalue (descending)
    if (wave == 1 && lane < M) {
#pragma unroll
        for (int i = 0; i < M; ++i) s.e[lane][i] = e[i];
    }
    double wi = 0.0;
    int rank = -1;
    if (tid < M) {
        auto diag_at = [&](int i) { return fa[((i >> 1) * H + (i >> 1)) * 4 + (i & 1) * 3]; };
        wi = diag_at(tid);
        rank = 0;
        for (int j = 0; j < M; ++j) {
            const double wj = diag_at(j);
            if (wj > wi || (wj == wi && j < tid)) ++rank;
        }
    }
    __syncthreads();
    if (rank >= 0) {
        s.w[rank] = wi;
        s.perm[rank] = tid;
    }
    __syncthreads();
}

// The general form, for a workgroup of NT threads (256, or 1 024: k_small_finish<1024>, round 5).  With 1 024
// threads only the independent work items of a step are spread wider -- three per thread instead of twelve at
// n = 64; everything whose ORDER matters (the convergence sums: per-thread partials over e = tid mod 256, wave
// shuffles, four wave totals) is done by the first 256 threads exactly as before: the same bits.
template <int NT>
__device__ void jacobi_eig_generic(jacobi_lds &s, int n) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = n + (n & 1);  // even; a padded index has zero row/column
    for (int e = tid; e < m * m; e += NT) {
        const int i = e / m, j = e - i * m;
        s.e[i][j] = i == j ? 1.0 : 0.0;
        if (i >= n || j >= n) s.a[i][j] = 0.0;
    }
    __syncthreads();
    const int half = m / 2;
    for (int sweep = 0; sweep < 30; ++sweep) {
        // convergence: off-diagonal mass against the diagonal
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < n * n && tid < 256; e += 256) {
            const int i = e / n, j = e - i * n;
            const double v = s.a[i][j];
            // dead directions carry -1e30 on the diagonal (k_small_rr): not part of the scale
            if (i == j) dia += v > -1e29 ? v * v : 0.0;
            else off += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_xor(off, o, 64);
            dia += __shfl_xor(dia, o, 64);
        }
        if (lane == 0 && wave < 4) {
            s.red[wave] = off;
            s.red[4 + wave] = dia;
        }
        __syncthreads();
        off = s.red[0] + s.red[1] + s.red[2] + s.red[3];
        dia = s.red[4] + s.red[5] + s.red[6] + s.red[7];
        __syncthreads();
        // off-diagonal mass at the rounding floor of an n x n matrix: (n * eps)^2 * ||diag||^2
        if (off <= 1.25e-32 * (double)(n * n) * dia || off == 0.0) break;

        for (int step = 0; step < m - 1; ++step) {
            if (tid < half) {
                // round-robin tournament: index m-1 stays, the others rotate
                int p, q;
                if (tid == 0) {
                    p = m - 1;
                    q = step;
                } else {
                    p = (step + tid) % (m - 1);
                    q = (step - tid + (m - 1)) % (m - 1);
                }
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                const double apq = s.a[p][q];
                double c = 1.0, sn = 0.0;
                if (apq != 0.0) {
                    const double th = (s.a[q][q] - s.a[p][p]) / (2.0 * apq);
                    const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    sn = t * c;
                }
                s.cs[tid][0] = c;
                s.cs[tid][1] = sn;
                s.pq[tid][0] = p;
                s.pq[tid][1] = q;
            }
            __syncthreads();
            // work items: half*half 2x2 blocks of A, then m*half column pairs of E
            const int nblk = half * half;
            for (int e = tid; e < nblk + m * half; e += NT) {
                if (e < nblk) {
                    const int kr = e / half, kc = e - kr * half;
                    const int p = s.pq[kr][0], q = s.pq[kr][1];
                    const int pc = s.pq[kc][0], qc = s.pq[kc][1];
                    const double c1 = s.cs[kr][0], s1 = s.cs[kr][1];
                    const double c2 = s.cs[kc][0], s2 = s.cs[kc][1];
                    const double a00 = s.a[p][pc], a01 = s.a[p][qc];
                    const double a10 = s.a[q][pc], a11 = s.a[q][qc];
                    // right: columns (pc, qc) <- (c2*x - s2*y, s2*x + c2*y)
                    const double t00 = c2 * a00 - s2 * a01, t01 = s2 * a00 + c2 * a01;
                    const double t10 = c2 * a10 - s2 * a11, t11 = s2 * a10 + c2 * a11;
                    // left: rows (p, q) <- (c1*rp - s1*rq, s1*rp + c1*rq)
                    double n00 = c1 * t00 - s1 * t10, n01 = c1 * t01 - s1 * t11;
                    double n10 = s1 * t00 + c1 * t10, n11 = s1 * t01 + c1 * t11;
                    if (kr == kc) {  // the rotated pair itself: annihilated exactly
                        n01 = 0.0;
                        n10 = 0.0;
                    }
                    s.a[p][pc] = n00;
                    s.a[p][qc] = n01;
                    s.a[q][pc] = n10;
                    s.a[q][qc] = n11;
                } else {
                    const int f = e - nblk;
                    const int r = f / half, kc = f - r * half;
                    const int pc = s.pq[kc][0], qc = s.pq[kc][1];
                    const double c2 = s.cs[kc][0], s2 = s.cs[kc][1];
                    const double ep = s.e[r][pc], eq = s.e[r][qc];
                    s.e[r][pc] = c2 * ep - s2 * eq;
                    s.e[r][qc] = s2 * ep + c2 * eq;
                }
            }
            __syncthreads();
        }
    }
    // sort descending (rank by counting)
    if (tid < n) {
        const double wi = s.a[tid][tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double wj = s.a[j][j];
            if (wj > wi || (wj == wi && j < tid)) ++rank;
        }
        s.w[rank] = wi;
        s.perm[rank] = tid;
    }
    __syncthreads();
}

__device__ void jacobi_eig(jacobi_lds &s, int n) {
    if (n == 12) return jacobi_eig_waves<12>(s);
    if (n == 8) return jacobi_eig_waves<8>(s);
    if (n == 4) return jacobi_eig_waves<4>(s);
    if (n <= 24) {
        jacobi_eig_fast(s, n);
        return;
    }
    jacobi_eig_generic<256>(s, n);
}

// Sum `nparts` per-workgroup partials of `nout` outputs (partial[p * nout + e]) in a fixed
// order with all 256 threads of the workgroup: eight interleaved slices of the partials are
// summed with the loads of a slice in flight together, then the slices are combined.
// out(e, value) is called by one thread per output; ends with a workgroup barrier.
template <int TL, typename F>
__device__ __forceinline__ void sum_partials(const double *__restrict__ partial, int nparts,
                                             int nout, double (*tmp)[TL], F out) {
    // tmp: [8][>= nout] doubles of LDS
    // (up to five outputs per thread at a time, sixteen loads of each in flight before the first
    // add: the kernel is one workgroup, nothing else hides the latency of a load, and the
    // partials come from the other XCDs' workgroups, i.e. from memory -- the 12 x 12 Rayleigh-Ritz
    // matrix, 1 152 slice sums, is then ONE round trip instead of ten)
    constexpr int IT = 5;
    const int total = nout * 8;
    for (int f0 = threadIdx.x; f0 < total; f0 += 256 * IT) {
        int sl[IT], el[IT];
        bool on[IT];
        double v[IT];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int f = f0 + 256 * j;
            on[j] = f < total;
            sl[j] = on[j] ? f / nout : 0;
            el[j] = on[j] ? f - sl[j] * nout : 0;
            v[j] = 0.0;
        }
        for (int base = 0; base < nparts; base += 128) {
            double x[IT][16];
#pragma unroll
            for (int j = 0; j < IT; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int p = base + sl[j] + 8 * k;
                    x[j][k] = (on[j] && p < nparts) ? partial[(int64_t)p * nout + el[j]] : 0.0;
                }
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int j = 0; j < IT; ++j) v[j] += x[j][k];
        }
#pragma unroll
        for (int j = 0; j < IT; ++j)
            if (on[j]) tmp[sl[j]][el[j]] = v[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nout; e += 256) {
        double v = tmp[0][e];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += tmp[k][e];
        out(e, v);
    }
    __syncthreads();
}

// plain eigen-decomposition of a global n x n matrix (debug entry point, dense path)
__global__ __launch_bounds__(256) void k_small_eig(const double *__restrict__ a, int n,
                                                    double *__restrict__ w, double *__restrict__ v) {
    __shared__ jacobi_lds s;
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        s.a[i][j] = 0.5 * (a[i * n + j] + a[j * n + i]);
    }
    __syncthreads();
    jacobi_eig(s, n);
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e / n, k = e - i * n;
        v[i * n + k] = s.e[i][s.perm[k]];
    }
    if (threadIdx.x < n) w[threadIdx.x] = s.w[threadIdx.x];
}

// ---------------------------------------------------------------------------
// dense path for 65 .. 128 vertices: one-sided (Hestenes) Jacobi in LDS
// ---------------------------------------------------------------------------
// The two-sided Jacobi above keeps the matrix AND the accumulated rotations in LDS: two
// (n + 1) x n arrays, 64 vertices at most.  Up to 128 vertices one array still fits the 160 KB,
// and the one-sided method needs no second one: rotating the COLUMNS of A = S + I (symmetric
// positive semi-definite: the eigenvalues of S lie in [-1, 1]) until they are mutually
// orthogonal turns column j into lambda_j v_j -- its norm is the eigenvalue (of S, plus 1), its
// direction the eigenvector.  A step orthogonalises m / 2 disjoint column pairs (round-robin
// tournament), four threads per pair: three dot products over the rows (partial sums by
// shuffles), one rotation, the two columns rewritten.  Recursion nodes of 65 .. 128 taxa are
// then ~1 ms of one workgroup instead of ~35 latency-bound LOBPCG iterations.
constexpr int DENSE2_MAX = 128;
constexpr int DENSE2_LD = DENSE2_MAX + 1;

// The sweeps: columns of a (m x m, m even) rotated pairwise until mutually orthogonal.  All 256
// threads of the workgroup call; returns false when 40 sweeps did not get there (never seen: the
// method converges quadratically; a caller reports it rather than hand back a half-rotated basis).
__device__ bool onesided_sweeps(double (*a)[DENSE2_LD], int m, int *s_rot) {
    const int tid = threadIdx.x;
    const int half = m / 2;
    const int pair = tid >> 2, part = tid & 3;
    bool done = false;
    for (int sweep = 0; sweep < 40 && !done; ++sweep) {
        if (tid == 0) *s_rot = 0;
        __syncthreads();
        for (int step = 0; step < m - 1; ++step) {
            int rotated = 0;
            if (pair < half) {
                // round-robin tournament: index m - 1 stays, the others rotate
                int p, q;
                if (pair == 0) {
                    p = m - 1;
                    q = step;
                } else {
                    p = (step + pair) % (m - 1);
                    q = (step - pair + (m - 1)) % (m - 1);
                }
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int r = part; r < m; r += 4) {
                    const double x = a[r][p], y = a[r][q];
                    al = fma(x, x, al);
                    be = fma(y, y, be);
                    ga = fma(x, y, ga);
                }
                al += __shfl_xor(al, 1, 64);
                be += __shfl_xor(be, 1, 64);
                ga += __shfl_xor(ga, 1, 64);
                al += __shfl_xor(al, 2, 64);
                be += __shfl_xor(be, 2, 64);
                ga += __shfl_xor(ga, 2, 64);
                // (the four threads of a pair hold the same sums: the same decision); columns
                // whose cosine is at the rounding floor of a 128-term dot product (~ n eps = 3e-14)
                // are orthogonal: rotating them again would never end
                // -- and a column whose squared norm is below 1e-24 IS zero: an eigenvalue -1 of S
                // (a bipartite component) is a null direction of A = S + I, its column shrinks to
                // rounding noise whose "cosine" with the other columns is anything, and the pair
                // would be rotated for ever (seen: a 2-tree bootstrap forest in the recursion sweep).
                // The wanted columns have norms near 2; what a 1e-12 column still overlaps with them
                // moves their eigenvalues by 1e-24.
                if (ga * ga > 1e-27 * al * be && al > 1e-24 && be > 1e-24) {
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
                    const double c = 1.0 / sqrt(fma(t, t, 1.0)), sn = c * t;
                    for (int r = part; r < m; r += 4) {
                        const double x = a[r][p], y = a[r][q];
                        a[r][p] = c * x - sn * y;
                        a[r][q] = sn * x + c * y;
                    }
                    rotated = 1;
                }
            }
            if (rotated) *s_rot = 1;  // (benign race: every writer stores 1)
            __syncthreads();
        }
        done = *s_rot == 0;
        __syncthreads();
    }
    return done;
}

// w3: the three largest eigenvalues of S and, in w3[3], 1.0 when the sweeps converged (0.0: they did not)
__global__ __launch_bounds__(256) void k_dense_onesided(const double *__restrict__ sd, int n,
                                                         double *__restrict__ w3,
                                                         double *__restrict__ v2) {
    extern __shared__ double dyn_lds[];
    double(*a)[DENSE2_LD] = (double(*)[DENSE2_LD])dyn_lds;  // a[row][col]
    __shared__ double s_norm[DENSE2_MAX];
    __shared__ int s_rot, s_top[3];
    const int tid = threadIdx.x;
    const int m = n + (n & 1);  // an odd n gets a zero column / row of padding
    for (int e = tid; e < m * m; e += 256) {
        const int i = e / m, j = e - i * m;
        double v = 0.0;
        if (i < n && j < n) v = i == j ? 1.0 : 0.5 * (sd[i * n + j] + sd[j * n + i]);
        a[i][j] = v;
    }
    __syncthreads();
    const bool ok = onesided_sweeps(a, m, &s_rot);
    // eigenvalues of S: column norms - 1; the three largest, the two leading unit columns
    if (tid < n) {
        double s2 = 0.0;
        for (int r = 0; r < n; ++r) s2 = fma(a[r][tid], a[r][tid], s2);
        s_norm[tid] = sqrt(s2);
    }
    __syncthreads();
    if (tid < n) {
        const double mine = s_norm[tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double o = s_norm[j];
            if (o > mine || (o == mine && j < tid)) ++rank;
        }
        if (rank < 3) s_top[rank] = tid;
    }
    __syncthreads();
    if (tid < 3) w3[tid] = tid < n ? s_norm[s_top[tid]] - 1.0 : 0.0;
    if (tid == 3) w3[3] = ok ? 1.0 : 0.0;
    for (int e = tid; e < 2 * n; e += 256) {
        const int k = e / n, r = e - k * n;
        const int col = s_top[k];
        v2[r * 2 + k] = a[r][col] / s_norm[col];
    }
}

// SVQB (Stathopoulos & Wu): from G = Y^T Y (k x k) build T (k x k) with
// (Y T)^T (Y T) = I on the kept directions; directions whose scaled eigenvalue is
// below drop_tol * largest are dropped (zero column of T, mask 0).
__global__ __launch_bounds__(256) void k_small_svqb(const double *__restrict__ g, int k,
                                                     double drop_tol, double *__restrict__ t,
                                                     int *__restrict__ mask) {
    __shared__ jacobi_lds s;
    __shared__ double dsc[MAXS];
    const int tid = threadIdx.x;
    if (tid < k) {
        const double d = g[tid * k + tid];
        dsc[tid] = d > 1e-290 ? 1.0 / sqrt(d) : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < k * k; e += 256) {
        const int i = e / k, j = e - i * k;
        double v = 0.5 * (g[i * k + j] + g[j * k + i]) * dsc[i] * dsc[j];
        if (dsc[i] == 0.0 || dsc[j] == 0.0) v = 0.0;  // dead column: decoupled, eigenvalue 0
        s.a[i][j] = v;
    }
    __syncthreads();
    jacobi_eig(s, k);
    const double wmax = s.w[0];
    for (int e = tid; e < k * k; e += 256) {
        const int i = e / k, c = e - i * k;
        const double lam = s.w[c];
        const bool keep = wmax > 0.0 && lam > drop_tol * wmax;
        t[i * k + c] = keep ? dsc[i] * s.e[i][s.perm[c]] / sqrt(lam) : 0.0;
    }
    if (tid < k) mask[tid] = (wmax > 0.0 && s.w[tid] > drop_tol * wmax) ? 1 : 0;
}

// Search-direction coefficients of the fused loop (b = 4, 8), one wave, registers and
// wave shuffles only: lane i holds row i of C (the top-b eigenvectors) and of D.
//   D <- the [P R] rows of C;  D <- (I - C C^T) D twice;  then classical Gram-Schmidt with
//   re-orthogonalisation over the columns in Ritz order, a column being dropped (zero,
//   mask 0) when less than drop_tol of its squared norm is left.
// Same role as the SVQB step of the general path below; cheaper because nothing iterates.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int B>
__device__ void rr_directions_wave(const jacobi_lds &s, int nq, double drop_tol,
                                   double *__restrict__ d_out, int *__restrict__ mask_p) {
    const int i = threadIdx.x;  // lane of wave 0
    const bool row = i < nq;
    double c[B], d[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const double v = row ? s.e[i][s.perm[k]] : 0.0;
        c[k] = v;
        d[k] = i >= B ? v : 0.0;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        double g[B][B];
#pragma unroll
        for (int a = 0; a < B; ++a)
#pragma unroll
            for (int k = 0; k < B; ++k) g[a][k] = row_sum<(3 * B + 15) / 16>(c[a] * d[k]);
#pragma unroll
        for (int k = 0; k < B; ++k)
#pragma unroll
            for (int a = 0; a < B; ++a) d[k] -= c[a] * g[a][k];
    }
    double n0[B];
#pragma unroll
    for (int k = 0; k < B; ++k) n0[k] = row_sum<(3 * B + 15) / 16>(d[k] * d[k]);
    int keep_mask = 0;
#pragma unroll
    for (int k = 0; k < B; ++k) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            double r[B];
#pragma unroll
            for (int j = 0; j < k; ++j) r[j] = row_sum<(3 * B + 15) / 16>(d[j] * d[k]);
#pragma unroll
            for (int j = 0; j < k; ++j) d[k] -= r[j] * d[j];
        }
        const double n1 = row_sum<(3 * B + 15) / 16>(d[k] * d[k]);
        const bool keep = n0[k] > 1e-290 && n1 > drop_tol * n0[k];
        const double sc = keep ? 1.0 / sqrt(n1) : 0.0;
        d[k] *= sc;
        keep_mask |= keep ? (1 << k) : 0;
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
        if (row) d_out[i * B + k] = d[k];
        if (i == k) mask_p[k] = (keep_mask >> k) & 1;
    }
}

// Rayleigh-Ritz: T (nq x nq) = Q^T A Q on an orthonormal basis Q = [X R P]; mask marks
// live basis columns.  Outputs
//   c (nq x b): top-b eigenvectors  -> X' = Q c,   theta[0..b) their values, theta[b] the next;
//   d (nq x b): coefficients of the next search directions P' = Q d, orthonormal and
//     orthogonal to X' *in coefficient space* (Q is orthonormal, so no tall Gram product is
//     needed): start from the [R P] part of c (the LOBPCG direction), project out c twice,
//     SVQB-orthonormalise with rank-revealing drop; mask_p[k] = 0 marks a dropped direction.
//   nparts > 0: tm holds nparts per-workgroup partials of T (k_gram_qaq), summed here in
//   fixed order.
// (mask, c, d, theta, mask_p may live in LDS: k_panel_rr_solve runs this in front of its panel pass)
// exact_from >= 0 (mixed-precision loop): only the products S Q_i of the columns i < exact_from (the X
// block) are exact -- S R came through the single-precision image of W, and S P is a combination that
// contains it.  Of the two entries of a pair (i < exact_from <= j) only the one with the exact product,
// Q_j^T (S Q_i), is used: the other carries the image's rounding as an ABSOLUTE error, on an entry (the
// coupling of X to a search direction) that is as small as the residual itself.  Inside the X block the
// entry with the product of the LOWER column is used likewise: the columns behind the wanted one take
// long steps along image products, their S X drifts by more, and the mean would rotate that drift into
// the wanted column (measured: a floor of 1.4e-13 ... 3e-13 under the residual of column 0).  Pairs of
// two search directions keep the mean: their entries are O(||S||).
__device__ __forceinline__ void small_rr_body(const double *__restrict__ tm, int nparts, int nq, int b,
                              const int *mask, double *c, double *d, double *theta, int *mask_p,
                              double drop_tol, int exact_from = -1) {
    __shared__ jacobi_lds s;
    __shared__ double cc[3 * MAXB][MAXB];
    __shared__ double dd[3 * MAXB][MAXB];
    __shared__ double gg[MAXB][MAXB];
    __shared__ double dsc[MAXB];
    const int tid = threadIdx.x;
    if (nparts > 0) {
        __shared__ double tmp[8][3 * MAXB * 3 * MAXB / 4];  // fused path: nq <= 24
        // parked in the eigenvector array until symmetrised
        sum_partials(tm, nparts, nq * nq, tmp, [&](int e, double v) { s.e[e / nq][e % nq] = v; });
    }
    for (int e = tid; e < nq * nq; e += 256) {
        const int i = e / nq, j = e - i * nq;
        double v = nparts > 0 ? 0.5 * (s.e[i][j] + s.e[j][i])
                              : 0.5 * (tm[i * nq + j] + tm[j * nq + i]);
        if (exact_from >= 0 && nparts > 0 && i != j && (i < exact_from || j < exact_from))
            v = i < j ? s.e[j][i] : s.e[i][j];
        const bool live = mask[i] && mask[j];
        if (!live) v = (i == j) ? -1e30 : 0.0;  // dead directions can never be selected
        s.a[i][j] = v;
    }
    __syncthreads();
    jacobi_eig(s, nq);
    for (int e = tid; e < nq * b; e += 256) {
        const int i = e / b, k = e - i * b;
        const double v = s.e[i][s.perm[k]];
        cc[i][k] = v;
        dd[i][k] = i >= b ? v : 0.0;
        c[e] = v;
    }
    if (tid <= b && tid < nq) theta[tid] = s.w[tid];
    __syncthreads();
    if (nq == b) return;  // plain rotation of X (start-up / refresh): no search directions
    if (nq == 3 * b && (b == 4 || b == 8)) {
        if (tid < 64) {
            if (b == 4) rr_directions_wave<4>(s, nq, drop_tol, d, mask_p);
            else rr_directions_wave<8>(s, nq, drop_tol, d, mask_p);
        }
        return;
    }
    // D <- (I - C C^T) D, twice
    for (int pass = 0; pass < 2; ++pass) {
        for (int e = tid; e < b * b; e += 256) {
            const int a = e / b, k = e - a * b;
            double g = 0.0;
            for (int i = 0; i < nq; ++i) g += cc[i][a] * dd[i][k];
            gg[a][k] = g;
        }
        __syncthreads();
        for (int e = tid; e < nq * b; e += 256) {
            const int i = e / b, k = e - i * b;
            double v = dd[i][k];
            for (int a = 0; a < b; ++a) v -= cc[i][a] * gg[a][k];
            dd[i][k] = v;
        }
        __syncthreads();
    }
    // SVQB on D (b columns in an nq-dimensional coefficient space)
    for (int e = tid; e < b * b; e += 256) {
        const int a = e / b, k = e - a * b;
        double g = 0.0;
        for (int i = 0; i < nq; ++i) g += dd[i][a] * dd[i][k];
        gg[a][k] = g;
    }
    __syncthreads();
    if (tid < b) dsc[tid] = gg[tid][tid] > 1e-290 ? 1.0 / sqrt(gg[tid][tid]) : 0.0;
    __syncthreads();
    for (int e = tid; e < b * b; e += 256) {
        const int a = e / b, k = e - a * b;
        s.a[a][k] = 0.5 * (gg[a][k] + gg[k][a]) * dsc[a] * dsc[k];
    }
    __syncthreads();
    jacobi_eig(s, b);
    const double wmax = s.w[0];
    // transform T = diag(dsc) U Lambda^-1/2 (dropped columns zero), D_hat = D T
    for (int e = tid; e < b * b; e += 256) {
        const int a = e / b, k = e - a * b;
        const double lam = s.w[k];
        const bool keep = wmax > 0.0 && lam > drop_tol * wmax;
        gg[a][k] = keep ? dsc[a] * s.e[a][s.perm[k]] / sqrt(lam) : 0.0;
    }
    if (tid < b) mask_p[tid] = (wmax > 0.0 && s.w[tid] > drop_tol * wmax) ? 1 : 0;
    __syncthreads();
    for (int e = tid; e < nq * b; e += 256) {
        const int i = e / b, k = e - i * b;
        double v = 0.0;
        for (int a = 0; a < b; ++a) v += dd[i][a] * gg[a][k];
        d[e] = v;
    }
}

__global__ __launch_bounds__(256) void k_small_rr(const double *__restrict__ tm, int nparts,
                                                   int nq, int b, const int *__restrict__ mask,
                                                   double *__restrict__ c, double *__restrict__ d,
                                                   double *__restrict__ theta,
                                                   int *__restrict__ mask_p, double drop_tol) {
    small_rr_body(tm, nparts, nq, b, mask, c, d, theta, mask_p, drop_tol);
}

// Projected SVQB (fused path, b <= 8).  Input: per-workgroup partials of
//   Cxp = [x p]^T r (2b x b),  G = r^T r (b x b),  cu = u^T r (1 x b)
// for an orthonormal [u x p].  With M = G - Cxp^T Cxp - cu^T cu (the Gram matrix of r
// after projecting [u x p] out) and M = D^-1 V L V^T D^-1 its scaled eigen-decomposition,
// T = D V L^-1/2 and K = -C T make   r T + [x p u] K   orthonormal and orthogonal to
// [u x p] up to the cancellation in M -- a second pass of the same step removes that.
// Output coef: rows [x (b) | p (b) | r (b) | u | 3 zero rows] x b, the operand of
// k_panel_tf; mask[c] = 0 marks a dropped direction (zero column).
// report (mapped host memory, may be null): [0, b) squared residual norms diag(G),
// [16, 16 + b] the current Ritz values, [40] the caller's sequence number, written last.
// (coef may live in LDS, mask may be null: k_panel_tf_solve runs this in front of its panel pass)
// (the working set is a template parameter: the general one below, or the compact one of a single block
// width -- same arithmetic, same bits)
struct orth_lds {
    jacobi_lds s;
    double cxp[16][8], gm[9][8], mm[8][8], tt[8][8], dsc[8];
    double tmp[8][3 * MAXB * 3 * MAXB / 4];
    __device__ void eig(int b) { jacobi_eig(s, b); }
};
template <int B>
struct orth_small_lds {
    jacobi_small_lds<B> s;
    double cxp[2 * B][B], gm[B + 1][B], mm[B][B], tt[B][B], dsc[B];
    double tmp[8][3 * B * B + B];
    double coef[PANEL_COEF_ROWS<B> * B];
    panel_red_t red;
    __device__ void eig(int) { jacobi_eig_waves<B>(s); }
};

template <typename ST>
__device__ __forceinline__ void small_orth_core(ST &L, const double *__restrict__ partial, int nparts, int b,
                                                double drop_tol, double *coef, int *mask,
                                                const double *__restrict__ theta, double *report, double seq) {
    auto &s = L.s;
    auto &cxp = L.cxp;
    auto &gm = L.gm;
    auto &mm = L.mm;
    auto &tt = L.tt;
    auto &dsc = L.dsc;
    const int tid = threadIdx.x;
    const int nxp = 2 * b * b, nout = 3 * b * b + b;
    sum_partials(partial, nparts, nout, L.tmp, [&](int e, double v) {
        if (e < nxp) cxp[e / b][e % b] = v;
        else gm[(e - nxp) / b][(e - nxp) % b] = v;
    });
    if (report) {
        // hand the norms to the host through mapped memory: data, system-scope fence, then the
        // sequence number the host is polling for (no stream event: recording one costs the
        // device a ~6 us bubble).  seq < 0: the caller sends the sequence number itself, later
        // (report_sequence): the fence then finds the stores long done instead of waiting a PCIe
        // round trip for them in front of the workgroup's rows.
        if (tid < b) report[tid] = gm[tid][tid];
        if (tid <= b) report[16 + tid] = theta[tid];
        if (seq >= 0.0) {
            __threadfence_system();
            __syncthreads();
            if (tid == 0) {
                ((volatile double *)report)[40] = seq;
                __threadfence_system();
            }
        }
    }
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e - i * b;
        double v = 0.5 * (gm[i][j] + gm[j][i]) - gm[b][i] * gm[b][j];
        for (int k = 0; k < 2 * b; ++k) v -= cxp[k][i] * cxp[k][j];
        mm[i][j] = v;
    }
    __syncthreads();
    if (tid < b) {
        // a column whose projected norm^2 is at the cancellation floor of G carries no
        // direction of its own: dead
        const double dgn = mm[tid][tid];
        dsc[tid] = (dgn > 1e-290 && dgn > 1e-13 * gm[tid][tid]) ? 1.0 / sqrt(dgn) : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, j = e - i * b;
        s.a[i][j] = 0.5 * (mm[i][j] + mm[j][i]) * dsc[i] * dsc[j];
    }
    __syncthreads();
    L.eig(b);
    const double wmax = s.w[0];
    for (int e = tid; e < b * b; e += 256) {
        const int i = e / b, c = e - i * b;
        const double lam = s.w[c];
        const bool keep = wmax > 0.0 && lam > drop_tol * wmax;
        tt[i][c] = keep ? dsc[i] * s.e[i][s.perm[c]] / sqrt(lam) : 0.0;
    }
    if (mask && tid < b) mask[tid] = (wmax > 0.0 && s.w[tid] > drop_tol * wmax) ? 1 : 0;
    __syncthreads();
    for (int e = tid; e < (3 * b + 4) * b; e += 256) {
        const int k = e / b, c = e - k * b;
        double v = 0.0;
        if (k < 2 * b) {
            for (int i = 0; i < b; ++i) v -= cxp[k][i] * tt[i][c];
        } else if (k < 3 * b) {
            v = tt[k - 2 * b][c];
        } else if (k == 3 * b) {
            for (int i = 0; i < b; ++i) v -= gm[b][i] * tt[i][c];
        }
        coef[e] = v;
    }
}

__device__ __forceinline__ void small_orth_body(const double *__restrict__ partial, int nparts, int b,
                                                double drop_tol, double *coef, int *mask,
                                                const double *__restrict__ theta, double *report, double seq) {
    __shared__ orth_lds L;
    small_orth_core(L, partial, nparts, b, drop_tol, coef, mask, theta, report, seq);
}

__global__ __launch_bounds__(256) void k_small_orth(const double *__restrict__ partial,
                                                     int nparts, int b, double drop_tol,
                                                     double *__restrict__ coef,
                                                     int *__restrict__ mask,
                                                     const double *__restrict__ theta,
                                                     double *report, double seq) {
    small_orth_body(partial, nparts, b, drop_tol, coef, mask, theta, report, seq);
}

// ---- the small solves in FRONT of the tall kernels that consume them
// A one-workgroup kernel between two tall ones costs its launch and the drain of the one before
// (~5 us a time, three times an iteration).  The tall kernel's workgroups can each run the small
// solve themselves, redundantly -- same partials, same code, same bits in every workgroup --
// and go straight on to their rows with the coefficients in LDS; workgroup 0 also publishes what
// later kernels and the host read (Ritz values, masks, the residual report).  Nothing is shared
// between the workgroups inside the kernel, so there is no release / acquire across XCDs (the
// variant that ran the solve in the TAIL of the producing kernel paid for exactly that,
// scs_panel.h).  Partials alternate between two buffers: a workgroup still summing the incoming
// ones must not see another one's outgoing ones; the search-direction mask alternates likewise.
template <int B>
__global__ __launch_bounds__(256) void k_panel_rr_solve(double *q, double *aq,
                                                         const double *__restrict__ u,
                                                         const double *__restrict__ tm, int nparts_in,
                                                         int solve, double *theta_g,
                                                         const int *__restrict__ maskp_in,
                                                         const int *__restrict__ mask_r,
                                                         int *maskp_out, double drop_tol, int n,
                                                         double *__restrict__ partial,
                                                         const double *__restrict__ dinv = nullptr,
                                                         double *__restrict__ zt = nullptr, int64_t ldz = 0,
                                                         int exact_from = -1) {
    __shared__ double c_s[3 * B * B], d_s[3 * B * B], th_s[B + 4];
    __shared__ int live_s[3 * B], mp_s[B];
    const int tid = threadIdx.x;
    if (solve) {
        if (tid < 3 * B) live_s[tid] = tid < B ? 1 : (tid < 2 * B ? maskp_in[tid - B] : mask_r[tid - 2 * B]);
        __syncthreads();
        small_rr_body(tm, nparts_in, 3 * B, B, live_s, c_s, d_s, th_s, mp_s, drop_tol, exact_from);
    } else {
        // first iteration, and after a refresh: X stays, no search directions
        for (int e = tid; e < 3 * B * B; e += 256) {
            c_s[e] = (e / B == e % B) ? 1.0 : 0.0;
            d_s[e] = 0.0;
        }
        if (tid <= B) th_s[tid] = theta_g[tid];
        if (tid < B) mp_s[tid] = 0;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (solve && tid <= B) theta_g[tid] = th_s[tid];
        if (tid < B) maskp_out[tid] = mp_s[tid];
    }
    panel_rr_body<B>(q, aq, u, c_s, d_s, th_s, n, partial, dinv, zt, ldz);
}

template <int B, bool GRAM, bool WRITE_Z>
__global__ __launch_bounds__(256) void k_panel_tf_solve(double *q, const double *__restrict__ u,
                                                         const double *__restrict__ part_in,
                                                         int nparts_in, double drop_tol, int *mask_r,
                                                         const double *__restrict__ theta_g,
                                                         double *report, double seq, int n,
                                                         double *__restrict__ partial,
                                                         const double *__restrict__ dinv,
                                                         double *__restrict__ zt, int64_t ldz) {
    __shared__ double coef_s[PANEL_COEF_ROWS<B> * B];
    const bool lead = blockIdx.x == 0;
    small_orth_body(part_in, nparts_in, B, drop_tol, coef_s, lead ? mask_r : nullptr, theta_g,
                    lead ? report : nullptr, -1.0);
    __syncthreads();
    panel_tf_body<B, GRAM, WRITE_Z>(q, u, coef_s, n, partial, dinv, zt, ldz);
    if (lead && report) {
        // the report's sequence number, after the rows (the end of the kernel makes it visible)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) ((volatile double *)report)[40] = seq;
    }
}

// Pass 2 of the orthonormalisation of R folded into the Gram kernel (scs_panel.h, round 5): the
// small solve on pass 1's Gram partials in front, then R and S R transformed on the way to Q^T S Q.
template <int B, int FINISH>
__global__ __launch_bounds__(256) void k_panel_gram_tf_solve(double *q, double *aq, const double *__restrict__ u,
                                                              const double *__restrict__ part_in, int nparts_in,
                                                              int *mask_r, const double *__restrict__ theta_g, int n,
                                                              const double *__restrict__ ypart, int nseg,
                                                              const double *__restrict__ dinv,
                                                              double *__restrict__ partial, int64_t chunk = 0,
                                                              const int32_t *__restrict__ splits = nullptr,
                                                              const double *__restrict__ coef1 = nullptr) {
    constexpr int NC = PANEL_COEF_ROWS<B> * B;
    __shared__ double coef_s[NC], coefa_s[NC];
    const bool lead = blockIdx.x == 0;
    small_orth_body(part_in, nparts_in, B, 0.0, coef_s, lead ? mask_r : nullptr, theta_g, nullptr, -1.0);
    __syncthreads();
    // coef1 (overlapped loop, k_symm_tri_tf): the operator was applied to the RAW residual block R~, and
    // pass 1 made R1 = R~ T1 + [x p u] K1 beside it.  With pass 2's R2 = R1 T2 + [x p u] K2:
    //   S R2 = (S R~) T1 T2 + S [x p u] (K1 T2 + K2)
    for (int e = threadIdx.x; e < NC; e += 256) {
        const int k = e / B, c = e - k * B;
        double v = coef_s[e];
        if (coef1) {
            if (k >= 2 * B && k < 3 * B) v = 0.0;
            for (int i = 0; i < B; ++i) v += coef1[k * B + i] * coef_s[(2 * B + i) * B + c];
        }
        coefa_s[e] = v;
    }
    __syncthreads();
    panel_gram_tf_body<B, FINISH>(q, aq, u, coef_s, coefa_s, n, ypart, nseg, dinv, partial, chunk, splits);
}

// Pass 1 of that orthonormalisation INSIDE the SYMM launch (round 5): the first n_tf workgroups run it
// (small solve on the compact LDS layout, then their share of the rows) while the others stream W against
// the raw residual block.  The two roles share the workgroup's LDS; nothing passes between them inside
// the launch.  Workgroup 0 publishes pass 1's coefficients for the Gram kernel (above).
template <int B, int CT, int RPW, int D, typename WT = double>
__global__ __launch_bounds__(256, 2) void k_symm_tri_tf(const WT *__restrict__ w, int64_t ld, int n,
                                                        const double *__restrict__ zt, int64_t ldz,
                                                        const int2 *__restrict__ tiles,
                                                        double *__restrict__ pdir, double *__restrict__ ptr_,
                                                        int n_tf, double *q, const double *__restrict__ u,
                                                        const double *__restrict__ part_in, int nparts_in,
                                                        double drop_tol, int *mask_r,
                                                        const double *__restrict__ theta_g, double *report,
                                                        double seq, double *__restrict__ partial,
                                                        double *__restrict__ coef_out) {
    union lds_t {
        symm_tri_lds<B, CT, WT> symm;
        orth_small_lds<B> tf;
    };
    __shared__ lds_t lds;
    const int bid = blockIdx.x;
    if (bid >= n_tf) {
        symm_tri_body<B, CT, RPW, D, WT>(w, ld, n, zt, ldz, tiles[bid - n_tf], pdir, ptr_, 0, lds.symm);
        return;
    }
    const bool lead = bid == 0;
    small_orth_core(lds.tf, part_in, nparts_in, B, drop_tol, lds.tf.coef, lead ? mask_r : nullptr, theta_g,
                    lead ? report : nullptr, -1.0);
    __syncthreads();
    if (lead)
        for (int e = threadIdx.x; e < PANEL_COEF_ROWS<B> * B; e += 256) coef_out[e] = lds.tf.coef[e];
    panel_tf_body<B, true, false>(q, u, lds.tf.coef, n, partial, nullptr, nullptr, 0, bid, n_tf, &lds.tf.red);
    if (lead && report) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) ((volatile double *)report)[40] = seq;
    }
}

// c = first b columns of the 3b x 3b identity, d = 0: the Rayleigh-Ritz coefficients that
// leave X alone and clear the search directions (first iteration, and after a refresh)
__global__ void k_unit_coeffs(double *__restrict__ c, double *__restrict__ d, int b) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 3 * b * b) return;
    c[e] = (e / b == e % b) ? 1.0 : 0.0;
    d[e] = 0.0;
}

// dense S for the small-V path: s[i][j] = dinv[i] * w[i][j] * dinv[j], zero diagonal
__global__ void k_dense_s(const double *__restrict__ w, int64_t ld, int n,
                          const double *__restrict__ dinv, double *__restrict__ s) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    s[idx] = i == j ? 0.0 : dinv[i] * w[(int64_t)i * ld + j] * dinv[j];
}

// ---------------------------------------------------------------------------
// matrix-free graphs (scs_matfree.h): creation, the operator, release
// ---------------------------------------------------------------------------
void scs_matfree_release(scs_graph *g) {
    mf_data *m = g->mf;
    if (!m) return;
    scs_dev_free(m->d_stack_off);
    scs_dev_free(m->st_val);
    scs_dev_free(m->st_sum);
    scs_dev_free(m->st_dep);
    scs_dev_free(m->sm_val);
    scs_dev_free(m->sm_sum);
    scs_dev_free(m->sm_dep);
    scs_dev_free(m->cy_pa);
    scs_dev_free(m->cy_ps);
    scs_dev_free(m->cy_dep);
    scs_dev_free(m->sm_cnt);
    scs_dev_free(m->sm_root);
    scs_dev_free(m->cy_cnt);
    scs_dev_free(m->x);
    scs_dev_free(m->y);
    scs_dev_free(m->slabs);
    delete m;
    g->mf = nullptr;
}

extern "C" int scs_graph_matrix_free(scs_ctx *ctx, const scs_tables *tb, int32_t max_block, scs_graph **out) {
    SCS_REQUIRE(ctx && tb && out, "scs_graph_matrix_free: null argument");
    SCS_REQUIRE(max_block == 4 || max_block == 8, "scs_graph_matrix_free: widest block 4 or 8 (asked: %d)", max_block);
    SCS_REQUIRE(ctx->comm.world == 1, "scs_graph_matrix_free: one device only (a measured comparison, DESIGN.md)");
    SCS_REQUIRE(tb->n_taxa > DENSE2_MAX, "scs_graph_matrix_free: more than %d taxa (the dense small paths need W)", DENSE2_MAX);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    SCS_TRY(scs_tables_finish(ctx, tb));
    hipStream_t s = ctx->stream;
    const int n = tb->n_taxa, M = tb->n_trees;
    auto *g = new scs_graph();
    g->n = n;
    g->row_begin = 0;
    g->row_end = n;
    g->ld = scs_round_up(n, SCS_LD_ALIGN);
    g->mf = new mf_data();
    mf_data *m = g->mf;
    m->tb = tb;
    m->b_cap = max_block;
    auto fail = [&](int rc) {
        scs_matfree_release(g);
        delete g;
        return rc;
    };
    // strips: a tree's stack never holds more entries than its deepest separator
    int32_t *d_maxd = nullptr;
    if (scs_dev_malloc((scs_ctx *)nullptr, (void **)&d_maxd, (size_t)M * 4) != hipSuccess) return fail(SCS_ENOMEM);
    hipMemsetAsync(d_maxd, 0, (size_t)M * 4, s);
    k_mf_maxdepth<<<(unsigned)(((tb->n_leaves + 63) / 64 + 255) / 256), 256, 0, s>>>(tb->d_tree_off, M, tb->d_adj_depth,
                                                                                     tb->n_leaves, d_maxd);
    std::vector<int32_t> maxd((size_t)M);
    hipError_t e = hipMemcpyAsync(maxd.data(), d_maxd, (size_t)M * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    scs_dev_free(d_maxd);
    if (e != hipSuccess) {
        scs_set_error("scs_graph_matrix_free: %s", hipGetErrorString(e));
        return fail(SCS_EHIP);
    }
    std::vector<int64_t> soff((size_t)M + 1, 0);
    for (int t = 0; t < M; ++t) soff[t + 1] = soff[t] + maxd[t] + 1;
    m->stack_total = soff[M];
    // chunks per direction: pieces of a few hundred leaves, at most 16
    m->chunks = std::max(1, std::min(16, tb->max_leaves / 256));
    const size_t strips = (size_t)2 * max_block * (size_t)m->chunks * (size_t)m->stack_total;
    const size_t slots = (size_t)2 * max_block * (size_t)m->chunks * (size_t)M;
    const size_t slab_bytes = (size_t)2 * M * (size_t)n * max_block * 8;
    if (scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->d_stack_off, (size_t)(M + 1) * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->st_val, std::max<size_t>(strips, 1) * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->st_sum, std::max<size_t>(strips, 1) * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->st_dep, std::max<size_t>(strips, 1) * 4) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->sm_val, std::max<size_t>(strips, 1) * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->sm_sum, std::max<size_t>(strips, 1) * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->sm_dep, std::max<size_t>(strips, 1) * 4) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->cy_pa, std::max<size_t>(strips, 1) * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->cy_ps, std::max<size_t>(strips, 1) * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->cy_dep, std::max<size_t>(strips, 1) * 4) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->sm_cnt, slots * 4) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->sm_root, slots * 4) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->cy_cnt, slots * 4) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->x, (size_t)n * max_block * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->y, (size_t)n * max_block * 8) != hipSuccess ||
        scs_dev_malloc((scs_ctx *)nullptr, (void **)&m->slabs, slab_bytes) != hipSuccess) {
        (void)hipGetLastError();
        scs_set_error("scs_graph_matrix_free: cannot allocate %.1f GB of slabs", slab_bytes / 1073741824.0);
        return fail(SCS_ENOMEM);
    }
    e = hipMemcpyAsync(m->d_stack_off, soff.data(), (size_t)(M + 1) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(m->slabs, 0, slab_bytes, s);  // taxa a tree does not hold stay zero
    if (e == hipSuccess) e = hipStreamSynchronize(s);                      // (`soff` goes out of scope)
    if (e != hipSuccess) {
        scs_set_error("scs_graph_matrix_free: %s", hipGetErrorString(e));
        return fail(SCS_EHIP);
    }
    *out = g;
    return SCS_OK;
}

// zt == null: the vector of ones (the degrees: column 0 of the result goes to g->d_deg)
int scs_matfree_apply(scs_ctx *ctx, scs_graph *g, const double *zt, int64_t ldz, int b, double *y_out,
                      hipStream_t s) {
    mf_data *m = g->mf;
    SCS_REQUIRE(m && (b == 4 || b == 8) && b <= m->b_cap, "matrix-free apply: block width %d (graph made for %d)", b,
                m ? m->b_cap : 0);
    const scs_tables *tb = m->tb;
    const int n = g->n, M = tb->n_trees;
    const int64_t nb = (int64_t)n * b;
    if (m->slabs_b != b) {
        // another width lays the slabs out differently: entries of taxa a tree does not hold must be zero
        SCS_HIP_CHECK(hipMemsetAsync(m->slabs, 0, (size_t)2 * M * (size_t)n * b * 8, s));
        m->slabs_b = b;
    }
    if (zt) {
        if (b == 4) k_mf_operand<4><<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(zt, ldz, n, m->x);
        else k_mf_operand<8><<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(zt, ldz, n, m->x);
    } else {
        k_mf_fill<<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(m->x, nb, 1.0);
    }
    mf_params a;
    a.tree_off = tb->d_tree_off;
    a.leaf_taxon = tb->d_leaf_taxon;
    a.adj_depth = tb->d_adj_depth;
    a.adj_val = tb->d_adj_val;
    a.tree_w = tb->d_tree_w;
    a.stack_off = m->d_stack_off;
    a.n_trees = M;
    a.n_taxa = n;
    a.stack_total = m->stack_total;
    a.st_val = m->st_val;
    a.st_sum = m->st_sum;
    a.st_dep = m->st_dep;
    a.x = m->x;
    a.y_slabs = m->slabs;
    a.chunks = m->chunks;
    a.sm_dep = m->sm_dep;
    a.sm_val = m->sm_val;
    a.sm_sum = m->sm_sum;
    a.cy_dep = m->cy_dep;
    a.cy_pa = m->cy_pa;
    a.cy_ps = m->cy_ps;
    a.sm_cnt = m->sm_cnt;
    a.sm_root = m->sm_root;
    a.cy_cnt = m->cy_cnt;
    const int64_t units = (int64_t)M * 2 * b;
    const int64_t threads = units * m->chunks;
    double *y = y_out ? y_out : m->y;
    // (strips are laid out for the graph's widest block; a narrower one uses a prefix of the units)
    if (b == 4) {
        k_mf_chunk<4, false><<<(unsigned)((threads + 63) / 64), 64, 0, s>>>(a);
        k_mf_carry<4><<<(unsigned)((units + 63) / 64), 64, 0, s>>>(a);
        k_mf_chunk<4, true><<<(unsigned)((threads + 63) / 64), 64, 0, s>>>(a);
        k_mf_reduce<4><<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(m->slabs, M, n, y);
    } else {
        k_mf_chunk<8, false><<<(unsigned)((threads + 63) / 64), 64, 0, s>>>(a);
        k_mf_carry<8><<<(unsigned)((units + 63) / 64), 64, 0, s>>>(a);
        k_mf_chunk<8, true><<<(unsigned)((threads + 63) / 64), 64, 0, s>>>(a);
        k_mf_reduce<8><<<(unsigned)((nb + 255) / 256), 256, 0, s>>>(m->slabs, M, n, y);
    }
    if (!zt) k_mf_column0<<<(n + 255) / 256, 256, 0, s>>>(y, n, b, g->d_deg);
    SCS_HIP_CHECK(hipGetLastError());
    ++m->n_apply;
    return SCS_OK;
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
namespace {

// a device block from the calling context's cache (scs_block_alloc): the entry points below
// name their context in t_ctx before the first allocation
thread_local scs_ctx *t_ctx = nullptr;

struct dbuf {
    void *p = nullptr;
    scs_ctx *owner = nullptr;
    ~dbuf() {
        if (p && owner) scs_block_release(owner, p);
        else if (p) scs_dev_free(p);
    }
    int alloc(size_t bytes) {
        if (t_ctx) {
            owner = t_ctx;
            return scs_block_alloc(t_ctx, bytes, &p);
        }
        SCS_HIP_CHECK(scs_dev_malloc((scs_ctx *)nullptr, &p, bytes ? bytes : 16));
        return SCS_OK;
    }
    double *d() const { return (double *)p; }
};

struct solver {
    scs_ctx *ctx = nullptr;
    scs_graph *g = nullptr;
    hipStream_t s = nullptr;
    int n = 0, b = 0, rows = 0, world = 1;
    bool use_mfma = true;
    dbuf q, aq, z, ypart, yloc, yfull, recv, u, part, part2, part3, small, splits_d;
    // round 5: the second orthonormalisation pass of R rides the Gram kernel behind the SYMM stream
    // (k_panel_gram_tf_solve; SCS_FOLD_PASS2=0 keeps it as a launch of its own in front of the stream)
    bool fold_pass2 = false;
    // ... and the first pass rides the SYMM launch itself (k_symm_tri_tf: one device, symmetric schedule;
    // SCS_OVERLAP_PASS1=0 keeps it in front)
    bool overlap_pass1 = false;
    struct {
        bool armed = false;
        double drop_tol = 0.0, seq = 0.0;
        double *report = nullptr;
    } tf1;
    dbuf coef1;
    // round 5, mixed precision: while use32 is set the symmetric SYMM streams the single-precision image
    // of W (g->d_w32, half the bytes; products and sums stay in double precision) -- the operator applied
    // to SEARCH DIRECTIONS only; S X and S P are renewed through W itself (scs_fiedler)
    bool use32 = false;
    bool gram_from32 = false;  // the Gram partials waiting for the next Rayleigh-Ritz solve hold image products
    int n_apply32 = 0;
    std::vector<char> ev32;  // per timed launch in `ev`: streamed the image
    const double *fold_u = nullptr;
    int *fold_mask_r = nullptr;
    const double *fold_theta = nullptr;
    size_t ypart_cap = 0;
    int64_t chunk = 0;
    int gram_blocks = 0;
    std::vector<int32_t> splits;
    // timing of SYMM launches
    std::vector<hipEvent_t> ev;
    std::vector<hipEvent_t> ev_ag;  // pairs around the timed all-gathers (multi-rank solves)
    hipEvent_t ev_cal[2] = {nullptr, nullptr};  // an empty pair (calibration)
    bool time_ag = false;           // set by fused_back for the launches it times
    int n_allgather = 0;
    int n_apply = 0;

    ~solver() {
        if (ctx) {
            ctx->event_pool.insert(ctx->event_pool.end(), ev.begin(), ev.end());
            ctx->event_pool.insert(ctx->event_pool.end(), ev_ag.begin(), ev_ag.end());
            for (auto e : ev_cal)
                if (e) ctx->event_pool.push_back(e);
        } else {
            for (auto e : ev_cal)
                if (e) hipEventDestroy(e);
            for (auto e : ev) hipEventDestroy(e);
            for (auto e : ev_ag) hipEventDestroy(e);
        }
    }
    int new_event(hipEvent_t *e) {
        if (ctx && !ctx->event_pool.empty()) {
            *e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
            return SCS_OK;
        }
        SCS_HIP_CHECK(hipEventCreate(e));
        return SCS_OK;
    }

    double *small_at(int off) const { return small.d() + off; }

    // the all-gather of an iteration, timed with its own event pair when fused_back times the launch
    int gather(const double *send, double *recvb, size_t count) {
        hipEvent_t a = nullptr, b = nullptr;
        if (time_ag) {
            SCS_TRY(new_event(&a));
            SCS_TRY(new_event(&b));
            ev_ag.push_back(a);
            ev_ag.push_back(b);
            SCS_HIP_CHECK(hipEventRecord(a, s));
        }
        SCS_TRY(scs_comm_allgather_f64(&ctx->comm, send, recvb, count, s));
        if (time_ag) SCS_HIP_CHECK(hipEventRecord(b, s));
        ++n_allgather;
        return SCS_OK;
    }

    int last_nseg = 1;  // column segments of the most recent k_symm launch

    // symmetric schedule (k_symm_tri): the whole matrix on this device, streamed by upper tiles;
    // or (SCS_BUILD_UPPER graphs, `part`) this rank's tiles of the job's upper triangle: its
    // product is then PARTIAL -- all V rows, to be added to the other ranks' in rank order
    bool tri = false, part_mode = false;
    int tri_ct = 2, tri_nct = 0, tri_ntiles = 0;
    int tri_rb_lo = 0, tri_rb_hi = 0x7FFFFFFF;  // this rank's row blocks (of TRI_TH rows)
    // The tile list is stored twice, forwards and backwards, and consecutive applications
    // alternate: the tiles an application streamed last are the ones the next one starts with,
    // so whatever part of W the 256 MiB Infinity Cache still holds is read there and not from
    // HBM (a cyclic sweep of more than the cache's size would find nothing).  Every tile writes
    // its own slabs of the partial sums: the order of the tiles does not touch the result.
    bool tri_backwards = false;
    dbuf tri_tiles, tri_pdir, tri_ptr, ysend;
    std::vector<int2> tiles_host;  // source of the asynchronous upload of the tile lists: lives as long as the solve
    // the image is streamed in 128 x 512 tiles (2 KB pieces of a row, as W's 256 doubles): its own tile list,
    // stored behind W's in tri_tiles; the partial-sum slabs are W's (fewer column tiles: a prefix)
    int tri32_nct = 0, tri32_ntiles = 0;
    double w32_bytes_per_apply = 0.0;
    int64_t ldz = 0;                 // leading dimension of the k-major operand z
    double w_bytes_per_apply = 0.0;  // bytes of W one application streams

    int launch_symm_tri(const double *zin, double *yout) {
        const int tw = tri_ct * 128;
        // global indices address the stored block: an SCS_BUILD_UPPER rank keeps rows from
        // row_begin and columns from col0 on; its transposed partials only need the slabs of
        // its own row blocks
        const double *w_eff = g->d_w - (int64_t)g->row_begin * g->ld - g->col0;
        double *ptr_eff = tri_ptr.d() - (int64_t)tri_rb_lo * n * b;
        // (tf1.armed: pass 1 of R's orthonormalisation in the first workgroups of the launch)
        const int n_tf = panel_blocks16();
        const int ntiles = use32 ? tri32_ntiles : tri_ntiles;
#define TRI_T(B_, CT_, RPW_, D_, WT_, W_)                                                            \
    if (tf1.armed)                                                                                   \
        k_symm_tri_tf<B_, CT_, RPW_, D_, WT_><<<ntiles + n_tf, 256, 0, s>>>(                         \
            W_, g->ld, n, zin, ldz, tile_list, tri_pdir.d(), ptr_eff, n_tf, q.d(), fold_u,           \
            part.d(), n_tf, tf1.drop_tol, fold_mask_r, fold_theta, tf1.report, tf1.seq, part2.d(),   \
            coef1.d());                                                                              \
    else                                                                                             \
        k_symm_tri<B_, CT_, RPW_, D_, WT_><<<ntiles, 256, 0, s>>>(W_, g->ld, n, zin, ldz,            \
                                                                      tile_list, tri_pdir.d(), ptr_eff)
#define TRI(B_, CT_, RPW_, D_) TRI_T(B_, CT_, RPW_, D_, double, w_eff)
        const int2 *tile_list = (const int2 *)tri_tiles.p + (use32 ? 2 * tri_ntiles : 0) + (tri_backwards ? ntiles : 0);
        tri_backwards = !tri_backwards;
        if (use32) {
            // (b = 4, 128 x 512 tiles: 8 % faster than 256-column ones, tools/symm_tri_bench.hip)
            TRI_T(4, 2, 2, 3, float, g->d_w32);
            ++n_apply32;
        } else if (b == 4) {
            if (tri_ct == 4) TRI(4, 4, 2, 3);
            else if (tri_ct == 1) TRI(4, 1, 4, 4);
            else TRI(4, 2, 4, 3);
        } else {
            if (tri_ct == 1) TRI(8, 1, 2, 6);
            else TRI(8, 2, 2, 4);
        }
#undef TRI
#undef TRI_T
        SCS_HIP_CHECK(hipGetLastError());
        tf1.armed = false;
        const bool img_tiles = use32;
        if (part_mode) {
            // this rank's partial product, unscaled, all V rows: gathered and added by the caller
            k_symm_tri_finish<<<(4 * n * b + 255) / 256, 256, 0, s>>>(tri_pdir.d(), ptr_eff, n, b, tw, tri_nct,
                                                                  nullptr, ysend.d(), tri_rb_lo, tri_rb_hi);
            SCS_HIP_CHECK(hipGetLastError());
            SCS_TRY(gather(ysend.d(), recv.d(), (size_t)n * b));
            if (yout) {
                k_sum_parts<<<(n * b + 255) / 256, 256, 0, s>>>(recv.d(), world, n, b, g->d_dinv, yout);
                SCS_HIP_CHECK(hipGetLastError());
            }
            last_nseg = world;
            return SCS_OK;
        }
        // one segment: scaled into yout, or unscaled into ypart for k_gram_qaq to fold in
        k_symm_tri_finish<<<(4 * n * b + 255) / 256, 256, 0, s>>>(tri_pdir.d(), tri_ptr.d(), n, b, img_tiles ? 512 : tw,
                                                              img_tiles ? tri32_nct : tri_nct,
                                                              yout ? g->d_dinv : nullptr,
                                                              yout ? yout : ypart.d());
        SCS_HIP_CHECK(hipGetLastError());
        last_nseg = 1;
        return SCS_OK;
    }

    // yout (rows x b) = dinv (.) (W_local * Z), Z given k-major in zin (b x ld);
    // yout == null leaves the column segments in ypart for the caller to combine
    int launch_symm(const double *zin, double *yout) {
        if (g->mf) {
            // matrix-free graph: one unscaled segment into ypart, as k_symm leaves its column segments
            SCS_TRY(scs_matfree_apply(ctx, g, zin, ldz, b, ypart.d(), s));
            last_nseg = 1;
            if (yout)
                k_symm_finish<<<(rows * b + 255) / 256, 256, 0, s>>>(ypart.d(), 1, rows, b, g->d_dinv,
                                                                     g->row_begin, yout);
            SCS_HIP_CHECK(hipGetLastError());
            return SCS_OK;
        }
        if (tri) return launch_symm_tri(zin, yout);
        const int64_t ld = g->ld;
        int rpw = 4, sdepth = 2;
        if (b == 16) rpw = 2;
        if (b == 4) sdepth = 4;
        const int rowblocks = (rows + 4 * rpw - 1) / (4 * rpw);
        const int n_macros = (int)(ld / (sdepth * SYMM_SUB));
        int nseg = (1024 + rowblocks - 1) / rowblocks;
        nseg = std::max(1, std::min(std::min(nseg, 4), n_macros));
        if (b == 8 && nseg < 2 && n_macros >= 2 && rowblocks < 2048) nseg = 2;
        const int mps = (n_macros + nseg - 1) / nseg;
        nseg = (n_macros + mps - 1) / mps;
        if (ypart_cap < (size_t)nseg * rows * b) {
            scs_set_error("internal: ypart buffer too small");
            return SCS_EINVAL;
        }
        dim3 grid((unsigned)rowblocks, (unsigned)nseg);
        switch (b) {
            case 4:
                if (use32) {
                    ++n_apply32;
                    k_symm<4, 4, 4, 3, float><<<grid, 256, 0, s>>>(g->d_w32, ld, rows, zin, ypart.d(), mps);
                } else {
                    k_symm<4, 4, 4, 3><<<grid, 256, 0, s>>>(g->d_w, ld, rows, zin, ypart.d(), mps);
                }
                break;
            case 8:
                k_symm<8, 4, 2, 2><<<grid, 256, 0, s>>>(g->d_w, ld, rows, zin, ypart.d(), mps);
                break;
            case 12:
                k_symm<12, 4, 2, 1><<<grid, 256, 0, s>>>(g->d_w, ld, rows, zin, ypart.d(), mps);
                break;
            case 16:
                k_symm<16, 2, 2, 2><<<grid, 256, 0, s>>>(g->d_w, ld, rows, zin, ypart.d(), mps);
                break;
            default:
                scs_set_error("unsupported block width %d (need 4, 8, 12 or 16)", b);
                return SCS_EUNSUP;
        }
        SCS_HIP_CHECK(hipGetLastError());
        last_nseg = nseg;
        if (yout)
            k_symm_finish<<<(rows * b + 255) / 256, 256, 0, s>>>(ypart.d(), nseg, rows, b,
                                                                 g->d_dinv, g->row_begin, yout);
        return SCS_OK;
    }

    int alloc_symm_buffers() {
        ypart_cap = (size_t)4 * rows * b;
        SCS_TRY(ypart.alloc(ypart_cap * 8));
        ldz = g->upper ? scs_round_up(n, SCS_LD_ALIGN) : g->ld;
        SCS_TRY(z.alloc((size_t)b * ldz * 8));
        SCS_HIP_CHECK(hipMemsetAsync(z.p, 0, (size_t)b * ldz * 8, s));
        w_bytes_per_apply = 8.0 * rows * (double)n;
        w32_bytes_per_apply = 4.0 * rows * (double)n;  // (k_symm on the image of a rank's rows; the symmetric schedule sets its own)
        if (g->mf)  // both sweeps read the leaf arrays (16 bytes a leaf); the slabs are written and read once
            w_bytes_per_apply = 2.0 * 16.0 * (double)g->mf->tb->n_leaves +
                                2.0 * 2.0 * 8.0 * (double)g->mf->tb->n_trees * (double)n * b;
        if (g->upper) {
            // the job's upper triangle, this rank's row blocks: TW = 256 (the diagonal tile of a
            // row block then starts where the build's stored cells of those rows start)
            if (b != 4 && b != 8) {
                scs_set_error("an SCS_BUILD_UPPER graph is solved with block width 4 or 8 (asked: %d)", b);
                return SCS_EUNSUP;
            }
            tri = true;
            part_mode = true;
            tri_ct = 2;
            const int tw = 256;
            tri_nct = (n + tw - 1) / tw;
            tri_rb_lo = g->row_begin / TRI_TH;
            tri_rb_hi = (g->row_end + TRI_TH - 1) / TRI_TH;
            std::vector<int2> tiles;
            for (int i = tri_rb_lo; i < tri_rb_hi; ++i)
                for (int j = i * TRI_TH / tw; j < tri_nct; ++j) tiles.push_back(make_int2(i, j));
            tri_ntiles = (int)tiles.size();
            tiles.insert(tiles.end(), tiles.rbegin(), tiles.rend());  // and backwards
            SCS_TRY(tri_tiles.alloc(std::max<size_t>(tiles.size(), 1) * sizeof(int2)));
            SCS_HIP_CHECK(hipMemcpyAsync(tri_tiles.p, tiles.data(), tiles.size() * sizeof(int2),
                                         hipMemcpyHostToDevice, s));
            SCS_HIP_CHECK(hipStreamSynchronize(s));  // `tiles` goes out of scope
            SCS_TRY(tri_pdir.alloc((size_t)tri_nct * n * b * 8));
            SCS_TRY(tri_ptr.alloc((size_t)(tri_rb_hi - tri_rb_lo) * n * b * 8));
            SCS_TRY(ysend.alloc((size_t)n * b * 8));
            if (!recv.p) SCS_TRY(recv.alloc((size_t)n * b * world * 8));
            w_bytes_per_apply = 8.0 * (double)tri_ntiles * TRI_TH * tw;
            return SCS_OK;
        }
        // W is symmetric: with all of it on this device only the tiles on and above the
        // diagonal need streaming (small matrices keep k_symm, whose column segments fill the
        // chip better).
        tri = !g->mf && world == 1 && rows == n && g->row_begin == 0 && n >= 4096 && (b == 4 || b == 8);
        if (tri) {
            const int tw = tri_ct * 128;
            const int n_rb = (n + TRI_TH - 1) / TRI_TH;
            tri_nct = (n + tw - 1) / tw;
            std::vector<int2> &tiles = tiles_host;
            tiles.clear();
            for (int i = 0; i < n_rb; ++i)
                for (int j = i * TRI_TH / tw; j < tri_nct; ++j) tiles.push_back(make_int2(i, j));
            tri_ntiles = (int)tiles.size();
            tiles.insert(tiles.end(), tiles.rbegin(), tiles.rend());  // and backwards
            {
                // the image's 128 x 512 tiles, behind W's list
                std::vector<int2> t32;
                tri32_nct = (n + 511) / 512;
                for (int i = 0; i < n_rb; ++i)
                    for (int j = i * TRI_TH / 512; j < tri32_nct; ++j) t32.push_back(make_int2(i, j));
                tri32_ntiles = (int)t32.size();
                tiles.insert(tiles.end(), t32.begin(), t32.end());
                tiles.insert(tiles.end(), t32.rbegin(), t32.rend());
                w32_bytes_per_apply = 4.0 * (double)tri32_ntiles * TRI_TH * 512;
            }
            SCS_TRY(tri_tiles.alloc(tiles.size() * sizeof(int2)));
            // (no wait here: the degree pass is still streaming W, and the list's source outlives the solve)
            SCS_HIP_CHECK(hipMemcpyAsync(tri_tiles.p, tiles.data(), tiles.size() * sizeof(int2),
                                         hipMemcpyHostToDevice, s));
            SCS_TRY(tri_pdir.alloc((size_t)std::max(tri_nct, tri32_nct) * n * b * 8));
            SCS_TRY(tri_ptr.alloc((size_t)n_rb * n * b * 8));
            w_bytes_per_apply = 8.0 * (double)tri_ntiles * TRI_TH * tw;
        }
        return SCS_OK;
    }

    // dst_panel[:, c0:c0+b] = S * src_panel[:, c0s:c0s+b]   (panels have ld 3b)
    int apply(const double *src, int c0s, double *dst, int c0d) {
        const int nb = n * b;
        k_scale_rows<<<(nb + 255) / 256, 256, 0, s>>>(src, 3 * b, c0s, b, n, g->d_dinv, z.d(),
                                                      ldz);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        SCS_TRY(new_event(&e0));
        SCS_TRY(new_event(&e1));
        ev.push_back(e0);
        ev.push_back(e1);
        ev32.push_back(use32 ? 1 : 0);
        SCS_HIP_CHECK(hipEventRecord(e0, s));
        SCS_TRY(launch_symm(z.d(), yloc.d()));
        SCS_HIP_CHECK(hipEventRecord(e1, s));
        ++n_apply;
        const double *yf = yloc.d();
        if (part_mode) {
            // (launch_symm_tri gathered the ranks' partial products and yloc holds their scaled
            // sum for all V rows)
        } else if (world > 1) {
            SCS_TRY(gather(yloc.d(), recv.d(), (size_t)chunk));
            k_unpack<<<(nb + 255) / 256, 256, 0, s>>>(recv.d(), chunk, b,
                                                      (const int32_t *)splits_d.p, world, n,
                                                      yfull.d());
            yf = yfull.d();
        }
        k_store_cols<<<(nb + 255) / 256, 256, 0, s>>>(yf, b, n, dst, 3 * b, c0d);
        return SCS_OK;
    }

    // out (ka x kb, device) = A^T B
    int gram(const double *a, int lda, int ka, const double *bm, int ldb, int kb, double *out,
             bool mfma) {
        const int nout = ka * kb;
        int nparts;
        if (mfma && ka <= 48 && kb <= 48) {
            nparts = std::min(gram_blocks, std::max(1, (n + 3) / 4));
            const int ta = (ka + 15) / 16, tb = (kb + 15) / 16;
#define GM(TA, TB)                                                                               \
    if (ta == TA && tb == TB)                                                                    \
        k_gram_mfma<TA, TB><<<nparts, 64, 0, s>>>(a, lda, ka, bm, ldb, kb, n, part.d());
            GM(1, 1) GM(1, 2) GM(1, 3) GM(2, 1) GM(2, 2) GM(2, 3) GM(3, 1) GM(3, 2) GM(3, 3)
#undef GM
        } else {
            nparts = std::min(gram_blocks, std::max(1, (n + GRAM_CH - 1) / GRAM_CH));
            k_gram<<<nparts, 256, 0, s>>>(a, lda, ka, bm, ldb, kb, n, part.d());
        }
        k_reduce_partials<<<(nout + 3) / 4, 256, 0, s>>>(part.d(), nparts, nout, out);
        SCS_HIP_CHECK(hipGetLastError());
        return SCS_OK;
    }

    int update(double *y, int ldy, int kc, double alpha, const double *a, int lda, int ka,
               const double *c, int ldc, double sign) {
        k_update<<<(n + 15) / 16, 256, 0, s>>>(y, ldy, kc, alpha, a, lda, ka, c, ldc, sign, n);
        SCS_HIP_CHECK(hipGetLastError());
        return SCS_OK;
    }

    // ---- fused iteration (scs_panel.h), block widths 4 and 8 ----
    static int panel_cap() {
        static const int cap = PANEL_BLOCKS_MAX;
        return cap;
    }
    int panel_blocks16() const { return std::max(1, std::min(panel_cap(), ((n + 15) / 16 + 3) / 4)); }
    int panel_blocks4() const { return std::max(1, std::min(panel_cap(), ((n + 3) / 4 + 3) / 4)); }

    template <int B>
    int fused_front(const double *uvec, const double *c, const double *d, const double *theta,
                    double *coef, int *mask_r, double drop_tol, double *report, double seq) {
        const int nb = panel_blocks16();
        k_panel_rr<B><<<nb, 256, 0, s>>>(q.d(), aq.d(), uvec, c, d, theta, n, part.d());
        k_small_orth<<<1, 256, 0, s>>>(part.d(), nb, B, drop_tol, coef, mask_r, theta, report, seq);
        k_panel_tf<B, true, false><<<nb, 256, 0, s>>>(q.d(), uvec, coef, n, part.d(), nullptr,
                                                      nullptr, 0);
        k_small_orth<<<1, 256, 0, s>>>(part.d(), nb, B, 0.0, coef, mask_r, theta, nullptr, 0.0);
        k_panel_tf<B, false, true><<<nb, 256, 0, s>>>(q.d(), uvec, coef, n, nullptr, g->d_dinv,
                                                      z.d(), ldz);
        SCS_HIP_CHECK(hipGetLastError());
        return SCS_OK;
    }

    // The same front with the small solves inside the tall kernels (k_panel_rr_solve,
    // k_panel_tf_solve): three launches instead of five, and the Rayleigh-Ritz kernel that used
    // to close the iteration is the head of the next one's first launch.  `solve` = 0: X stays
    // and the search directions are cleared (first iteration, after a refresh).
    template <int B>
    int fused_front_solve(const double *uvec, int nparts_in, int solve, double *theta,
                          const int *maskp_in, int *mask_r, int *maskp_out, double drop_tol,
                          double *report, double seq) {
        const int nb = panel_blocks16();
        if (overlap_pass1) {
            // the Rayleigh-Ritz kernel hands the SYMM the scaled RAW residual block; pass 1 is armed for
            // the SYMM launch of fused_back.  (One-sided Gram entries, exact_from = B, in this loop always:
            // S R is composed from S applied to the raw block and the products of what pass 1 projects out,
            // a difference of terms up to 1e6 times its size when a column's residual lies almost inside
            // [X P] -- its rounding is then what the image's is, small_rr_body.  Found by
            // tools/fuzz_mixed_precision.py: with the mean of both entries the all-double loop left a
            // converged column's noise grow until the block blew up, graphs with isolated vertices.)
            k_panel_rr_solve<B><<<nb, 256, 0, s>>>(q.d(), aq.d(), uvec, part3.d(), nparts_in, solve, theta,
                                                   maskp_in, mask_r, maskp_out, drop_tol, n, part.d(),
                                                   g->d_dinv, z.d(), ldz, B);
            SCS_HIP_CHECK(hipGetLastError());
            tf1.armed = true;
            tf1.drop_tol = drop_tol;
            tf1.report = report;
            tf1.seq = seq;
            return SCS_OK;
        }
        if (fold_pass2) {
            // round 5: pass 1 writes Z; pass 2 rides the Gram kernel behind the SYMM stream (fused_back)
            k_panel_rr_solve<B><<<nb, 256, 0, s>>>(q.d(), aq.d(), uvec, part3.d(), nparts_in, solve, theta,
                                                   maskp_in, mask_r, maskp_out, drop_tol, n, part.d(), nullptr,
                                                   nullptr, 0, gram_from32 ? B : -1);
            k_panel_tf_solve<B, true, true><<<nb, 256, 0, s>>>(q.d(), uvec, part.d(), nb, drop_tol, mask_r,
                                                               theta, report, seq, n, part2.d(), g->d_dinv,
                                                               z.d(), ldz);
            SCS_HIP_CHECK(hipGetLastError());
            return SCS_OK;
        }
        k_panel_rr_solve<B><<<nb, 256, 0, s>>>(q.d(), aq.d(), uvec, part2.d(), nparts_in, solve, theta,
                                               maskp_in, mask_r, maskp_out, drop_tol, n, part.d(), nullptr,
                                               nullptr, 0, gram_from32 ? B : -1);
        k_panel_tf_solve<B, true, false><<<nb, 256, 0, s>>>(q.d(), uvec, part.d(), nb, drop_tol, mask_r,
                                                            theta, report, seq, n, part2.d(), nullptr,
                                                            nullptr, 0);
        k_panel_tf_solve<B, false, true><<<nb, 256, 0, s>>>(q.d(), uvec, part2.d(), nb, 0.0, mask_r, theta,
                                                            nullptr, 0.0, n, nullptr, g->d_dinv, z.d(),
                                                            ldz);
        SCS_HIP_CHECK(hipGetLastError());
        return SCS_OK;
    }

    // AR = S R (R's scaled transpose already sits in z) into AQ's R slot, then the
    // partials of Q^T AQ (into pout); returns their count.
    // (Folding k_symm_tri_finish into k_gram_qaq -- every lane of the Gram kernel sums the ~60
    // tile partials of one AR entry, same order, same bits, 32 loads in flight -- was measured:
    // 25.1 us against 7.7 + 7.8 for the two launches: 128 workgroups cannot hide the latency of
    // that many dependent rounds of loads; the finish kernel spreads them over 625.)
    template <int B>
    int fused_back(int *nparts, double *pout) {
        // k_symm is timed with HIP events on every fourth launch only: recording an event costs
        // the device a ~6 us bubble
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool timed = (n_apply & 3) == 0;
        time_ag = timed && world > 1;
        if (timed) {
            SCS_TRY(new_event(&e0));
            SCS_TRY(new_event(&e1));
            ev.push_back(e0);
            ev.push_back(e1);
            ev32.push_back(use32 ? 1 : 0);
        }
        gram_from32 = use32;
        const int nb = fold_pass2 ? panel_blocks16() : panel_blocks4();
        const int nb16 = panel_blocks16();  // partials of pass 1's Gram products (k_panel_tf_solve)
        *nparts = nb;
        if (part_mode) {
            // the gathered partial products (world x V x b, unscaled) are added in rank order and
            // scaled inside the Gram kernel
            if (timed) SCS_HIP_CHECK(hipEventRecord(e0, s));
            SCS_TRY(launch_symm(z.d(), nullptr));
            if (timed) SCS_HIP_CHECK(hipEventRecord(e1, s));
            ++n_apply;
            if (fold_pass2)
                k_panel_gram_tf_solve<B, 3><<<nb, 256, 0, s>>>(q.d(), aq.d(), fold_u, part2.d(), nb16, fold_mask_r,
                                                               fold_theta, n, recv.d(), world, g->d_dinv, pout);
            else
                k_gram_qaq<B, 3><<<nb, 256, 0, s>>>(q.d(), aq.d(), n, recv.d(), world, g->d_dinv, pout);
        } else if (world == 1) {
            if (timed) SCS_HIP_CHECK(hipEventRecord(e0, s));
            SCS_TRY(launch_symm(z.d(), nullptr));
            if (timed) SCS_HIP_CHECK(hipEventRecord(e1, s));
            ++n_apply;
            if (fold_pass2)
                k_panel_gram_tf_solve<B, 1><<<nb, 256, 0, s>>>(q.d(), aq.d(), fold_u, part2.d(), nb16, fold_mask_r,
                                                               fold_theta, n, ypart.d(), last_nseg, g->d_dinv, pout,
                                                               0, nullptr, overlap_pass1 ? coef1.d() : nullptr);
            else
                k_gram_qaq<B, 1><<<nb, 256, 0, s>>>(q.d(), aq.d(), n, ypart.d(), last_nseg,
                                                       g->d_dinv, pout);
        } else {
            if (timed) SCS_HIP_CHECK(hipEventRecord(e0, s));
            SCS_TRY(launch_symm(z.d(), yloc.d()));
            if (timed) SCS_HIP_CHECK(hipEventRecord(e1, s));
            ++n_apply;
            SCS_TRY(gather(yloc.d(), recv.d(), (size_t)chunk));
            // (the gathered slices go straight into AQ's R slot inside the Gram kernel)
            if (fold_pass2)
                k_panel_gram_tf_solve<B, 2><<<nb, 256, 0, s>>>(q.d(), aq.d(), fold_u, part2.d(), nb16, fold_mask_r,
                                                               fold_theta, n, recv.d(), world, g->d_dinv, pout,
                                                               (int64_t)chunk, (const int32_t *)splits_d.p);
            else
                k_gram_qaq<B, 2><<<nb, 256, 0, s>>>(q.d(), aq.d(), n, recv.d(), world, g->d_dinv, pout,
                                                    (int64_t)chunk, (const int32_t *)splits_d.p);
        }
        time_ag = false;
        SCS_HIP_CHECK(hipGetLastError());
        return SCS_OK;
    }
};

// layout of the small-matrix scratch (doubles)
constexpr int SM_G = 0;                         // up to 48 x 48
constexpr int SM_T = SM_G + 48 * 48;            // svqb transform (<= 16 x 16) / rr coeffs (48 x 16)
constexpr int SM_C = SM_T + 48 * 16;            // projection coefficients (<= 32 x 16)
constexpr int SM_D = SM_C + 48 * 16;            // search-direction coefficients (48 x 16)
constexpr int SM_THETA = SM_D + 48 * 16;        // theta[0..b], padded
constexpr int SM_RN = SM_THETA + 32;            // residual norms^2 [b]
constexpr int SM_MASK = SM_RN + 32;             // int mask[48] (as 24 doubles)
constexpr int SM_TOTAL = SM_MASK + 32;

void sign_flip_and_store(const std::vector<double> &col0, const std::vector<double> &col1, int n,
                         double *maps) {
    // reference: sklearn/utils/extmath.py _deterministic_vector_sign_flip
    for (int c = 0; c < 2; ++c) {
        const std::vector<double> &v = c == 0 ? col0 : col1;
        int arg = 0;
        double best = -1.0;
        for (int i = 0; i < n; ++i)
            if (std::fabs(v[i]) > best) {
                best = std::fabs(v[i]);
                arg = i;
            }
        const double sg = v[arg] < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < n; ++i) maps[(size_t)i * 2 + c] = sg * v[i];
    }
}

}  // namespace

static int gather_full_rows(scs_ctx *ctx, scs_graph *g, std::vector<int32_t> &splits) {
    return scs_gather_row_splits(ctx, g->row_begin, g->row_end, g->n, splits);
}

// small-V path: dense S on the device, full Jacobi, top two eigenvectors
// 65 .. 128 vertices: dense S on the device, one-sided Jacobi in LDS (k_dense_onesided)
static int fiedler_dense_onesided(scs_ctx *ctx, scs_graph *g, double *maps, scs_stats *st) {
    const int n = g->n;
    hipStream_t s = ctx->stream;
    SCS_REQUIRE(ctx->comm.world == 1 && !g->upper, "dense path needs the whole matrix on one rank (V = %d)", n);
    dbuf sd, wv, vv;
    SCS_TRY(sd.alloc((size_t)n * n * 8));
    SCS_TRY(wv.alloc(4 * 8));
    SCS_TRY(vv.alloc((size_t)n * 2 * 8));
    k_dense_s<<<(n * n + 255) / 256, 256, 0, s>>>(g->d_w, g->ld, n, g->d_dinv, sd.d());
    const size_t lds = (size_t)(n + (n & 1)) * DENSE2_LD * sizeof(double);
    // (the attribute is per function AND device: set on every call -- a process may drive several
    // devices and threads, and a flag set once would cover only the first)
    SCS_HIP_CHECK(hipFuncSetAttribute((const void *)k_dense_onesided, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      DENSE2_MAX * DENSE2_LD * (int)sizeof(double)));
    k_dense_onesided<<<1, 256, lds, s>>>(sd.d(), n, wv.d(), vv.d());
    SCS_HIP_CHECK(hipGetLastError());
    std::vector<double> w(4), v((size_t)n * 2), dinv(n);
    SCS_HIP_CHECK(hipMemcpyAsync(w.data(), wv.p, 4 * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipMemcpyAsync(v.data(), vv.p, (size_t)n * 2 * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipMemcpyAsync(dinv.data(), g->d_dinv, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<double> c0(n), c1(n);
    for (int i = 0; i < n; ++i) {
        c0[i] = v[(size_t)i * 2 + 0] * dinv[i];
        c1[i] = v[(size_t)i * 2 + 1] * dinv[i];
    }
    sign_flip_and_store(c0, c1, n, maps);
    if (st) {
        st->block = 0;
        st->converged = w[3] != 0.0 ? 1 : 0;
        st->lambda[0] = w[0];
        st->lambda[1] = w[1];
        st->lambda_next = w[2];
    }
    if (w[3] == 0.0) {
        scs_set_error("scs_fiedler: the dense one-sided Jacobi did not converge in 40 sweeps (V = %d)", n);
        return SCS_ENOCONV;
    }
    return SCS_OK;
}

static int fiedler_dense(scs_ctx *ctx, scs_graph *g, double *maps, scs_stats *st) {
    const int n = g->n;
    hipStream_t s = ctx->stream;
    SCS_REQUIRE(ctx->comm.world == 1, "dense small-V path needs world == 1 (V = %d)", n);
    dbuf sd, wv, vv;
    SCS_TRY(sd.alloc((size_t)n * n * 8));
    SCS_TRY(wv.alloc((size_t)n * 8));
    SCS_TRY(vv.alloc((size_t)n * n * 8));
    k_dense_s<<<(n * n + 255) / 256, 256, 0, s>>>(g->d_w, g->ld, n, g->d_dinv, sd.d());
    k_small_eig<<<1, 256, 0, s>>>(sd.d(), n, wv.d(), vv.d());
    std::vector<double> w(n), v((size_t)n * n), dinv(n);
    SCS_HIP_CHECK(hipMemcpyAsync(w.data(), wv.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipMemcpyAsync(v.data(), vv.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipMemcpyAsync(dinv.data(), g->d_dinv, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<double> c0(n), c1(n);
    for (int i = 0; i < n; ++i) {
        c0[i] = v[(size_t)i * n + 0] * dinv[i];
        c1[i] = (n > 1 ? v[(size_t)i * n + 1] : 0.0) * dinv[i];
    }
    sign_flip_and_store(c0, c1, n, maps);
    if (st) {
        st->block = 0;
        st->converged = 1;
        st->lambda[0] = w[0];
        st->lambda[1] = n > 1 ? w[1] : 0.0;
        st->lambda_next = n > 2 ? w[2] : 0.0;
    }
    return SCS_OK;
}

extern "C" int scs_fiedler(scs_ctx *ctx, scs_graph *g, const double *x_init, double tol,
                           int32_t max_iter, int32_t block, double *maps_out, scs_stats *stats) {
    SCS_REQUIRE(ctx && g && maps_out, "scs_fiedler: null argument");
    SCS_REQUIRE(g->n >= 2, "scs_fiedler: need at least 2 vertices (have %d)", g->n);
    SCS_REQUIRE(!g->upper || g->n > MAXS, "scs_fiedler: an SCS_BUILD_UPPER graph needs more than %d vertices", MAXS);
    SCS_REQUIRE(block >= 0 && block <= MAXB, "scs_fiedler: block must be in [0, %d]", MAXB);
    SCS_REQUIRE(tol > 0.0 && max_iter >= 1, "scs_fiedler: tol must be > 0 and max_iter >= 1");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    t_ctx = ctx;
    hipStream_t s = ctx->stream;
    const int n = g->n;
    scs_stats st_local;
    scs_stats *st = stats ? stats : &st_local;
    memset(st, 0, sizeof(*st));
    st->n_vertices = n;

    // timing events live in the context (created once: this call is made thousands of times
    // per recursion)
    for (auto &e : ctx->solve_events)
        if (!e) SCS_HIP_CHECK(hipEventCreate(&e));
    hipEvent_t ev_a = ctx->solve_events[0], ev_b = ctx->solve_events[1];
    SCS_HIP_CHECK(hipEventRecord(ev_a, s));
    const auto t_entry = std::chrono::steady_clock::now();

    if (n <= DENSE2_MAX) SCS_TRY(scs_graph_prepare_degrees(ctx, g));

    // 65 .. 96 vertices on one rank: the one-sided dense solve.  Its time grows with n^2 (n - 1
    // steps per sweep, each rewriting two columns per pair): measured 1.9 ms at 80 vertices
    // against 3.0 ms for LOBPCG, 4.5 against 3.5 ms at 120 -- the crossover is near 100.
    // An asked-for block width keeps the iterative path.
    static const int dense_max = 96;
    if (n > MAXS && n <= dense_max && block == 0 && ctx->comm.world == 1 && !g->upper) {
        SCS_TRY(fiedler_dense_onesided(ctx, g, maps_out, st));
        SCS_HIP_CHECK(hipEventRecord(ev_b, s));
        SCS_HIP_CHECK(hipEventSynchronize(ev_b));
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev_a, ev_b);
        st->solve_ms = ms;
        return SCS_OK;
    }
    if (n <= MAXS) {
        SCS_TRY(fiedler_dense(ctx, g, maps_out, st));
        SCS_HIP_CHECK(hipEventRecord(ev_b, s));
        SCS_HIP_CHECK(hipEventSynchronize(ev_b));
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev_a, ev_b);
        st->solve_ms = ms;
        return SCS_OK;
    }

    // ---- mixed precision (round 5; SCS_LOWP=0 off, 1, 2 = default).  The loop applies the operator to
    // its SEARCH DIRECTIONS only (S X and S P are carried by linear updates), and a search direction does
    // not need S to sixteen digits: with the symmetric schedule at width 4 the SYMM streams a single-
    // precision image of W -- half the bytes, products and sums still in double precision -- that the
    // degree pass writes on its way through W.  What the image's rounding leaves in S X and S P is removed
    // by renewing both through W itself (two applications) when the residual passes 1e-8, and
    // once more should it stop halving; mode 1 goes on in double precision after the first renewal, mode 2
    // stays with the image.  The confirmation at the end always applies W, and a solve it sends back into the loop
    // continues without the image.
    const int lowp_mode = getenv("SCS_LOWP") ? atoi(getenv("SCS_LOWP")) : 2;
    const double lowp_tol = 1e-8, lowp_tol2 = 0.0;
    // (only the default loop hands its Rayleigh-Ritz solve the one-sided entries: small_rr_body)
    const bool loop_fused = !(scs_dbg("SCS_LEGACY_LOOP") && atoi(scs_dbg("SCS_LEGACY_LOOP"))) &&
                            !(scs_dbg("SCS_SPLIT_SMALL") && atoi(scs_dbg("SCS_SPLIT_SMALL"))) &&
                            !(scs_dbg("SCS_FOLD_PASS2") && !atoi(scs_dbg("SCS_FOLD_PASS2")));
    // (an image above SCS_LOWP_MAX_BYTES, default 16 GiB -- about 65 000 vertices -- is not made: what it
    // saves a solve of that size, some 80 ms, is less than what an allocation of tens of GB can cost)
    const double lowp_max_bytes = getenv("SCS_LOWP_MAX_BYTES") ? atof(getenv("SCS_LOWP_MAX_BYTES")) : 16.0 * (1u << 30);
    // (round 6: a row-partitioned job -- whole rows on every rank, k_symm -- keeps the image of each rank's rows:
    // the first real multi-GPU run streams per rank what the one-GPU run streams, not twice that)
    const bool image_rows = ctx->comm.world > 1 && !g->upper;
    const bool image_one = ctx->comm.world == 1 && !g->upper && g->row_begin == 0 && g->row_end == n;
    const bool image_ok = !g->mf && lowp_mode > 0 && n >= 4096 && loop_fused && (image_one || image_rows) &&
                          (g->have_w32 || 4.0 * (double)(g->row_end - g->row_begin) * (double)g->ld <= lowp_max_bytes);
    // default width: 4 while the panel kernels and the 3b x 3b Rayleigh-Ritz solve weigh
    // against the SYMM stream, 8 once streaming W dominates (measured crossover ~ 16k) -- unless the
    // loop can stream the single-precision image, which exists at width 4: half the bytes an
    // application outweigh the iterations width 8 saves (configs[3], 50 000 vertices: 48 iterations and
    // 78 ms against 35 and 101)
    int b = block ? block : ((n >= 16384 && !image_ok) ? 8 : 4);
    if (g->upper && b > 8) b = 8;  // the symmetric SYMM kernel comes in widths 4 and 8
    if (g->mf) b = std::min(b <= 4 ? 4 : 8, g->mf->b_cap);  // (the sweep kernel likewise)
    {
        const int allowed[] = {16, 12, 8, 4};
        const int cap = (n - 2) / 3;  // 3b basis vectors + the constraint must fit in V
        int pick = 4;
        for (int a : allowed)
            if (a <= b && a <= cap) {
                pick = a;
                break;
            }
        b = pick;  // (>= 4: more than the one or two pairs wanted)
    }
    const int q3 = 3 * b;

    const bool want32 = image_ok && b == 4;
    // (the iterative path allocates and clears its buffers while k_degrees streams W, and only then waits
    // for the degrees)
    SCS_TRY(scs_graph_prepare_degrees_begin(ctx, g, want32));

    solver sv;
    sv.ctx = ctx;
    sv.g = g;
    sv.s = s;
    sv.n = n;
    sv.b = b;
    sv.rows = g->row_end - g->row_begin;
    sv.world = ctx->comm.world;
    sv.use_mfma = !(scs_dbg("SCS_NO_MFMA") && atoi(scs_dbg("SCS_NO_MFMA")));
    sv.gram_blocks = 256;
    SCS_TRY(gather_full_rows(ctx, g, sv.splits));
    int max_rows = 0;
    for (int r = 0; r < sv.world; ++r)
        max_rows = std::max(max_rows, sv.splits[r + 1] - sv.splits[r]);
    sv.chunk = g->upper ? (int64_t)n * b : (int64_t)max_rows * b;  // (upper: whole partial products travel)

    SCS_TRY(sv.q.alloc((size_t)n * q3 * 8));
    SCS_TRY(sv.aq.alloc((size_t)n * q3 * 8));
    SCS_TRY(sv.alloc_symm_buffers());
    SCS_TRY(sv.yloc.alloc((size_t)sv.chunk * 8));
    SCS_TRY(sv.u.alloc((size_t)n * 8));
    SCS_TRY(sv.part.alloc((size_t)1024 * q3 * q3 * 8));
    SCS_TRY(sv.part2.alloc((size_t)1024 * q3 * q3 * 8));
    SCS_TRY(sv.part3.alloc((size_t)1024 * q3 * q3 * 8));
    SCS_TRY(sv.small.alloc((size_t)SM_TOTAL * 8));
    if (sv.world > 1 && !sv.part_mode) {
        SCS_TRY(sv.yfull.alloc((size_t)n * b * 8));
        SCS_TRY(sv.recv.alloc((size_t)sv.chunk * sv.world * 8));
        SCS_TRY(sv.splits_d.alloc((size_t)(sv.world + 1) * 4));
        SCS_HIP_CHECK(hipMemcpyAsync(sv.splits_d.p, sv.splits.data(), (size_t)(sv.world + 1) * 4,
                                     hipMemcpyHostToDevice, s));
    }
    SCS_HIP_CHECK(hipMemsetAsync(sv.q.p, 0, (size_t)n * q3 * 8, s));
    SCS_HIP_CHECK(hipMemsetAsync(sv.aq.p, 0, (size_t)n * q3 * 8, s));
    SCS_HIP_CHECK(hipMemsetAsync(sv.yloc.p, 0, (size_t)sv.chunk * 8, s));
    SCS_HIP_CHECK(hipMemsetAsync(sv.small.p, 0, (size_t)SM_TOTAL * 8, s));

    double *Q = sv.q.d(), *AQ = sv.aq.d();
    // panel layout [X | P | R]: the blocks R is projected against are contiguous
    double *X = Q, *R = Q + 2 * b;
    double *AX = AQ;
    double *G = sv.small_at(SM_G), *T = sv.small_at(SM_T), *C = sv.small_at(SM_C);
    double *D = sv.small_at(SM_D);
    double *TH = sv.small_at(SM_THETA), *RN = sv.small_at(SM_RN);
    int *MASK = (int *)sv.small_at(SM_MASK);
    const double drop_tol = 1e-13;

    SCS_TRY(scs_graph_prepare_degrees(ctx, g));  // (begun at the top: the degrees are needed from here on)
    sv.use32 = want32 && g->have_w32 && (sv.tri ? sv.tri_ct == 2 : (image_rows && !sv.part_mode));
    scs_loop_policy policy;  // the stop / renew / confirm rules (scs_policy.h)
    policy.tol = tol;
    policy.lowp_mode = lowp_mode;
    policy.lowp_tol = lowp_tol;
    policy.lowp_tol2 = lowp_tol2;
    policy.lowp_state = sv.use32 ? 1 : 0;  // image in use: 1 + the renewals of S X / S P made so far; 0: off
    const bool constrained = g->n_isolated == 0;
    const int want = constrained ? 1 : 2;

    // constraint vector
    dbuf x0d;
    if (x_init) {
        SCS_TRY(x0d.alloc((size_t)n * 8));
        SCS_HIP_CHECK(hipMemcpyAsync(x0d.p, x_init, (size_t)n * 8, hipMemcpyHostToDevice, s));
    }
    if (constrained)
        k_trivial<<<(n + 255) / 256, 256, 0, s>>>(g->d_deg, 1.0 / g->dd_norm, n, sv.u.d());

    // project the constraint out of a panel block (twice) and SVQB-orthonormalise it;
    // companion (may be null) receives the same column transform
    auto project_u = [&](double *Y, int passes) -> int {
        if (!constrained) return SCS_OK;
        for (int pass = 0; pass < passes; ++pass) {
            SCS_TRY(sv.gram(sv.u.d(), 1, 1, Y, q3, b, C, false));
            SCS_TRY(sv.update(Y, q3, b, 1.0, sv.u.d(), 1, 1, C, b, -1.0));
        }
        return SCS_OK;
    };
    auto svqb = [&](double *Y, double *AY, int *mask_out) -> int {
        for (int pass = 0; pass < 2; ++pass) {
            SCS_TRY(sv.gram(Y, q3, b, Y, q3, b, G, sv.use_mfma));
            k_small_svqb<<<1, 256, 0, s>>>(G, b, pass == 0 ? drop_tol : 0.0, T, mask_out);
            // second pass: every surviving direction is kept (dead columns stay dead: dsc = 0)
            SCS_TRY(sv.update(Y, q3, b, 0.0, Y, q3, b, T, b, 1.0));
            if (AY) SCS_TRY(sv.update(AY, q3, b, 0.0, AY, q3, b, T, b, 1.0));
        }
        return SCS_OK;
    };
    // mask handling: pass 2 would overwrite mask with all-live for surviving columns and 0
    // for dead ones (their Gram diagonal is 0 -> dsc 0 -> eigenvalue 0 -> dropped since
    // 0 > 0 is false), so the mask of the second pass is the final one.

    // ---- start block
    k_init_block<<<(n * b + 255) / 256, 256, 0, s>>>(Q, b, n, x_init ? x0d.d() : nullptr);
    SCS_TRY(project_u(X, 2));
    SCS_TRY(svqb(X, nullptr, MASK));
    SCS_TRY(sv.apply(Q, 0, AQ, 0));
    // rotate X so that X^T A X is diagonal
    SCS_TRY(sv.gram(X, q3, b, AX, q3, b, G, sv.use_mfma));
    {
        k_fill_int<<<1, 64, 0, s>>>(MASK, 48, 1);
        k_small_rr<<<1, 256, 0, s>>>(G, 0, b, b, MASK, T, D, TH, MASK + b, drop_tol);
        SCS_TRY(sv.update(X, q3, b, 0.0, X, q3, b, T, b, 1.0));
        SCS_TRY(sv.update(AX, q3, b, 0.0, AX, q3, b, T, b, 1.0));
        k_fill_int<<<1, 64, 0, s>>>(MASK + b, b, 0);  // no search directions yet
    }

    const int res_blocks = (n + 255) / 256;
    dbuf res_part;
    SCS_TRY(res_part.alloc((size_t)res_blocks * b * 8));
    std::vector<double> h_rn(b), h_th(b + 1);
    int iter = 0;
    bool converged = false;
    double final_res[2] = {0.0, 0.0};

    // Fused iteration (scs_panel.h) for the default block widths; SCS_LEGACY_LOOP=1 keeps
    // the one-kernel-per-step formulation (also used for widths 12 and 16).
    const bool fused = (b == 4 || b == 8) &&
                       !(scs_dbg("SCS_LEGACY_LOOP") && atoi(scs_dbg("SCS_LEGACY_LOOP")));
    if (fused) {
        if (!ctx->h_report) {
            SCS_HIP_CHECK(hipHostMalloc((void **)&ctx->h_report, 64 * sizeof(double),
                                        hipHostMallocMapped | hipHostMallocCoherent));
            SCS_HIP_CHECK(hipHostGetDevicePointer((void **)&ctx->d_report, ctx->h_report, 0));
            memset(ctx->h_report, 0, 64 * sizeof(double));
        }
        k_unit_coeffs<<<(3 * b * b + 255) / 256, 256, 0, s>>>(T, D, b);
    }
    const double *uvec = constrained ? sv.u.d() : nullptr;
    // SCS_SPLIT_SMALL=1: the small solves as one-workgroup kernels of their own (the round-3 loop)
    const bool split_small = scs_dbg("SCS_SPLIT_SMALL") && atoi(scs_dbg("SCS_SPLIT_SMALL"));
    sv.fold_pass2 = fused && !split_small && !(scs_dbg("SCS_FOLD_PASS2") && !atoi(scs_dbg("SCS_FOLD_PASS2")));
    sv.overlap_pass1 = sv.fold_pass2 && sv.tri && !sv.part_mode && sv.world == 1 &&
                       !(scs_dbg("SCS_OVERLAP_PASS1") && !atoi(scs_dbg("SCS_OVERLAP_PASS1")));
    if (sv.overlap_pass1) SCS_TRY(sv.coef1.alloc((size_t)(3 * 8 + 4) * 8 * 8));
    sv.fold_u = uvec;
    sv.fold_mask_r = MASK + 2 * b;
    sv.fold_theta = TH;
    double *qaq_out = sv.fold_pass2 ? sv.part3.d() : sv.part2.d();
    int rr_solve = 0, rr_parts = 0, mpar = 0;
    int *maskp[2] = {MASK + b, MASK + 32};

    for (iter = 0; iter < max_iter; ++iter) {
        if (fused) {
            // The whole iteration is enqueued before the host looks at the residual norms the
            // first small kernel reported: the device never waits for the host.  Nothing
            // after that kernel touches X, so on a stop X is the block the norms belong to.
            int nparts = 0;
            const double seq = (double)(++ctx->report_seq);
            if (!split_small) {
                // (the Rayleigh-Ritz solve on the previous iteration's Gram partials opens the
                // first launch)
                if (b == 4) {
                    SCS_TRY(sv.fused_front_solve<4>(uvec, rr_parts, rr_solve, TH, maskp[mpar], MASK + 2 * b,
                                                    maskp[mpar ^ 1], drop_tol, ctx->d_report, seq));
                    SCS_TRY(sv.fused_back<4>(&nparts, qaq_out));
                } else {
                    SCS_TRY(sv.fused_front_solve<8>(uvec, rr_parts, rr_solve, TH, maskp[mpar], MASK + 2 * b,
                                                    maskp[mpar ^ 1], drop_tol, ctx->d_report, seq));
                    SCS_TRY(sv.fused_back<8>(&nparts, qaq_out));
                }
                rr_parts = nparts;
                rr_solve = 1;
                mpar ^= 1;
            } else {
                if (b == 4) {
                    SCS_TRY(sv.fused_front<4>(uvec, T, D, TH, C, MASK + 2 * b, drop_tol, ctx->d_report, seq));
                    SCS_TRY(sv.fused_back<4>(&nparts, sv.part.d()));
                } else {
                    SCS_TRY(sv.fused_front<8>(uvec, T, D, TH, C, MASK + 2 * b, drop_tol, ctx->d_report, seq));
                    SCS_TRY(sv.fused_back<8>(&nparts, sv.part.d()));
                }
                k_small_rr<<<1, 256, 0, s>>>(sv.part.d(), nparts, q3, b, MASK, T, D, TH, MASK + b,
                                             drop_tol);
            }
            SCS_HIP_CHECK(hipGetLastError());
            {
                // wait for this iteration's report (the device is already working on the rest of
                // the iteration); give up if the stream died
                volatile double *hr = (volatile double *)ctx->h_report;
                const auto t_wait = std::chrono::steady_clock::now();
                unsigned spins = 0;
                while (hr[40] != seq) {
                    __builtin_ia32_pause();
                    // (a report normally arrives within a few microseconds; a host thread that has
                    // waited much longer lets others run -- eight ranks may share the cores)
                    if (spins > (1u << 18) && (spins & 0x3FF) == 0) sched_yield();
                    if ((++spins & 0xFFFF) == 0) {
                        const hipError_t qe = hipStreamQuery(s);
                        if (qe != hipSuccess && qe != hipErrorNotReady) {
                            scs_set_error("scs_fiedler: stream failed: %s", hipGetErrorString(qe));
                            return SCS_EHIP;
                        }
                        if (qe == hipSuccess && hr[40] != seq) {
                            // everything ran and the report never arrived
                            if (std::chrono::steady_clock::now() - t_wait > std::chrono::seconds(5)) {
                                scs_set_error("scs_fiedler: residual report lost at iteration %d", iter);
                                return SCS_EHIP;
                            }
                        }
                    }
                }
            }
            for (int j = 0; j < b; ++j) h_rn[j] = ((volatile double *)ctx->h_report)[j];
            for (int j = 0; j <= b; ++j) h_th[j] = ((volatile double *)ctx->h_report)[16 + j];
        } else {
            // residual block and its norms
            k_residual<<<res_blocks, 256, 0, s>>>(Q, AQ, b, TH, n, res_part.d());
            k_reduce_partials<<<(b + 3) / 4, 256, 0, s>>>(res_part.d(), res_blocks, b, RN);
            SCS_HIP_CHECK(hipMemcpyAsync(h_rn.data(), RN, (size_t)b * 8, hipMemcpyDeviceToHost, s));
            SCS_HIP_CHECK(hipMemcpyAsync(h_th.data(), TH, (size_t)(b + 1) * 8, hipMemcpyDeviceToHost, s));
            SCS_HIP_CHECK(hipStreamSynchronize(s));
        }
        double worst = 0.0;
        for (int j = 0; j < want; ++j) worst = std::max(worst, std::sqrt(h_rn[j]));
        for (int j = 0; j < want; ++j) final_res[constrained ? 1 : j] = std::sqrt(h_rn[j]);
        static const bool trace_res = scs_dbg("SCS_TRACE_RESIDUAL") && atoi(scs_dbg("SCS_TRACE_RESIDUAL"));
        if (trace_res) fprintf(stderr, "[lobpcg] it %3d  residual %.3e  theta %.12g %.12g\n", iter, worst, h_th[0], h_th[1]);
        if (!(worst == worst)) {
            scs_set_error("scs_fiedler: NaN residual at iteration %d", iter);
            return SCS_EHIP;
        }
        const int action = policy.step(worst);
        const bool stop = action == scs_loop_policy::STOP;
        if (action == scs_loop_policy::RENEW) {
            // S X and S P anew through W itself, behind the iteration already enqueued (its Gram matrix,
            // formed with the old products, steers one more Rayleigh-Ritz step: coefficients only).
            // What the image adds to S X afterwards is its rounding (~1e-10 ||S||) times the steps still
            // to be taken, which are of the size of the residual over the spectral gap: one renewal clears
            // what the long early steps left (measured: without it 15 more iterations); a second one is
            // made when the residual stops halving (four iterations) (a second threshold on the residual itself exists and is off).
            // After the FIRST renewal a loop that has not halved its residual in eight iterations stops for the
            // confirmation (the image's rounding has become the floor); before it LOBPCG has plateaus of its own
            // -- ten iterations at 4e-5 on the `bootstrap` workload -- five orders above the image's rounding.
            sv.use32 = false;
            SCS_TRY(sv.apply(Q, 0, AQ, 0));
            SCS_TRY(sv.apply(Q, b, AQ, b));
            sv.use32 = policy.image_in_use();
            ++st->lowp_renewals;
        }
        if (stop) {
            sv.use32 = false;  // the confirmation, and whatever follows it, through W
            // confirm against a freshly applied operator (AX drifts by linear updates)
            if (policy.begin_confirmation()) {
                SCS_TRY(sv.apply(Q, 0, AQ, 0));
                SCS_TRY(sv.gram(X, q3, b, AX, q3, b, G, sv.use_mfma));
                k_fill_int<<<1, 64, 0, s>>>(MASK, 48, 1);
                k_small_rr<<<1, 256, 0, s>>>(G, 0, b, b, MASK, T, D, TH, MASK + b, drop_tol);
                SCS_TRY(sv.update(X, q3, b, 0.0, X, q3, b, T, b, 1.0));
                SCS_TRY(sv.update(AX, q3, b, 0.0, AX, q3, b, T, b, 1.0));
                // restart the search directions after the refresh: P slot dead and zero
                k_fill_int<<<1, 64, 0, s>>>(MASK + b, b, 0);
                k_residual<<<res_blocks, 256, 0, s>>>(Q, AQ, b, TH, n, res_part.d());
                k_reduce_partials<<<(b + 3) / 4, 256, 0, s>>>(res_part.d(), res_blocks, b, RN);
                SCS_HIP_CHECK(hipMemcpyAsync(h_rn.data(), RN, (size_t)b * 8, hipMemcpyDeviceToHost, s));
                SCS_HIP_CHECK(hipMemcpyAsync(h_th.data(), TH, (size_t)(b + 1) * 8,
                                             hipMemcpyDeviceToHost, s));
                SCS_HIP_CHECK(hipStreamSynchronize(s));
                double w2 = 0.0;
                for (int j = 0; j < want; ++j) w2 = std::max(w2, std::sqrt(h_rn[j]));
                for (int j = 0; j < want; ++j) final_res[constrained ? 1 : j] = std::sqrt(h_rn[j]);
                bool conv = false;
                if (policy.confirm(w2, &conv)) {
                    converged = conv;
                    break;
                }
                if (fused) k_unit_coeffs<<<(3 * b * b + 255) / 256, 256, 0, s>>>(T, D, b);
                rr_solve = 0;
            } else {
                converged = worst <= tol;
                break;
            }
        }

        if (fused) continue;
        // R <- orthonormal complement of [u, X, P] within span(R); P is orthonormal and
        // orthogonal to X by construction (k_small_rr), dead P columns are zero vectors
        // S u = u exactly, so R = S X - X theta inherits X's orthogonality to u; one
        // pass keeps rounding drift from accumulating
        SCS_TRY(project_u(R, 1));
        for (int pass = 0; pass < 2; ++pass) {
            SCS_TRY(sv.gram(Q, q3, 2 * b, R, q3, b, C, sv.use_mfma));
            SCS_TRY(sv.update(R, q3, b, 1.0, Q, q3, 2 * b, C, b, -1.0));
        }
        SCS_TRY(svqb(R, nullptr, MASK + 2 * b));
        SCS_TRY(sv.apply(Q, 2 * b, AQ, 2 * b));

        // Rayleigh-Ritz on the orthonormal basis [X P R]; new X and new P in one pass each
        SCS_TRY(sv.gram(Q, q3, q3, AQ, q3, q3, G, sv.use_mfma));
        k_small_rr<<<1, 256, 0, s>>>(G, 0, q3, b, MASK, T, D, TH, MASK + b, drop_tol);
        k_rr_update<<<(n + 15) / 16, 256, 0, s>>>(Q, b, q3, T, D, n);
        k_rr_update<<<(n + 15) / 16, 256, 0, s>>>(AQ, b, q3, T, D, n);
        SCS_HIP_CHECK(hipGetLastError());
    }

    // what an event pair around a launch adds to the launch's own time on this stream (the pair's two records
    // back to back, the stream busy with the solve's last kernels): reported beside the timed launches
    hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr;
    if (sv.new_event(&ev_c0) == SCS_OK && sv.new_event(&ev_c1) == SCS_OK) {
        sv.ev_cal[0] = ev_c0;
        sv.ev_cal[1] = ev_c1;
        hipEventRecord(ev_c0, s);
        hipEventRecord(ev_c1, s);
    }

    // ---- results: the two wanted columns, scaled, from the device (not the whole panel)
    dbuf maps_d;
    SCS_TRY(maps_d.alloc((size_t)n * 2 * 8));
    k_extract_maps<<<(n + 255) / 256, 256, 0, s>>>(Q, q3, n, g->d_dinv, 1.0 / g->dd_norm, constrained ? 1 : 0,
                                                  maps_d.d());
    // (page-locked staging: a pageable destination goes through the runtime's own bounce buffers)
    struct pinned_buf {
        scs_ctx *ctx;
        void *p = nullptr;
        ~pinned_buf() {
            if (p) scs_pinned_release(ctx, p);
        }
    } cols_h{ctx};
    SCS_TRY(scs_pinned_get(ctx, (size_t)n * 2 * 8, &cols_h.p));
    const double *cols = (const double *)cols_h.p;
    SCS_HIP_CHECK(hipMemcpyAsync(cols_h.p, maps_d.p, (size_t)n * 2 * 8, hipMemcpyDeviceToHost, s));
    // (fused loop: TH already holds the next iteration's Ritz values; h_th has X's)
    if (!fused)
        SCS_HIP_CHECK(hipMemcpyAsync(h_th.data(), TH, (size_t)(b + 1) * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipEventRecord(ev_b, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<double> c0(n), c1(n);
    for (int i = 0; i < n; ++i) {
        c0[i] = cols[(size_t)i * 2];
        c1[i] = cols[(size_t)i * 2 + 1];
    }
    if (constrained) {
        st->lambda[0] = 1.0;
        st->lambda[1] = h_th[0];
        st->lambda_next = b > 1 ? h_th[1] : 0.0;
    } else {
        st->lambda[0] = h_th[0];
        st->lambda[1] = h_th[1];
        st->lambda_next = b > 2 ? h_th[2] : 0.0;
    }
    sign_flip_and_store(c0, c1, n, maps_out);

    st->block = b;
    st->iterations = iter;
    st->n_apply = sv.n_apply;
    st->n_apply32 = sv.n_apply32;
    st->converged = converged ? 1 : 0;
    st->used_constraint = constrained ? 1 : 0;
    st->resid[0] = final_res[0];
    st->resid[1] = final_res[1];
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev_a, ev_b);
    st->solve_ms = ms;
    double tot = 0.0, mn = 1e300, tot32 = 0.0;
    int n_timed = 0, n_timed32 = 0;
    for (size_t i = 0; i + 1 < sv.ev.size(); i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, sv.ev[i], sv.ev[i + 1]) == hipSuccess) {
            if (i / 2 < sv.ev32.size() && sv.ev32[i / 2]) {
                tot32 += t;
                ++n_timed32;
                continue;
            }
            tot += t;
            mn = std::min(mn, (double)t);
            ++n_timed;
        }
    }
    // the fused loop times every fourth launch: scale the sample to all launches (of its kind: the
    // launches that streamed the single-precision image are reported on their own)
    if (n_timed > 0) tot *= (double)(sv.n_apply - sv.n_apply32) / n_timed;
    if (n_timed32 > 0) tot32 *= (double)sv.n_apply32 / n_timed32;
    st->apply_ms_total = tot;
    st->apply32_ms_total = tot32;
    st->apply32_bytes = sv.n_apply32 ? sv.w32_bytes_per_apply + 8.0 * (double)n * b + 8.0 * (double)sv.rows * b : 0.0;
    st->apply_ms_min = n_timed ? mn : 0.0;
    {
        float t = 0.f;
        st->event_pair_ms = (sv.ev_cal[1] && hipEventElapsedTime(&t, sv.ev_cal[0], sv.ev_cal[1]) == hipSuccess) ? t : 0.0;
    }
    // W bytes one application streams (all of this rank's rows, or the upper tiles of the
    // symmetric schedule) + the block in and out
    st->apply_bytes = sv.w_bytes_per_apply + 8.0 * (double)n * b + 8.0 * (double)sv.rows * b;
    {
        // the timed all-gathers (every fourth iteration of the fused loop), scaled to all of them
        double ag = 0.0;
        int n_ag = 0;
        for (size_t i = 0; i + 1 < sv.ev_ag.size(); i += 2) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, sv.ev_ag[i], sv.ev_ag[i + 1]) == hipSuccess) {
                ag += t;
                ++n_ag;
            }
        }
        st->n_allgather = sv.n_allgather;
        st->allgather_ms_total = n_ag ? ag * (double)sv.n_allgather / n_ag : 0.0;
        // bytes one all-gather delivers to this rank: `world` chunks of `chunk` doubles
        st->allgather_bytes = sv.world > 1 ? 8.0 * (double)sv.chunk * sv.world : 0.0;
    }
    if (n >= 4096 && scs_dbg("SCS_TRACE_SOLVES") && atoi(scs_dbg("SCS_TRACE_SOLVES")))
        fprintf(stderr, "[solve] V %d block %d iterations %d applies %d image %d renewals %d refreshes %d solve_ms %.3f "
                        "wall_ms %.3f residual %.3e gap %.3e\n", n, b, iter, sv.n_apply, sv.n_apply32, st->lowp_renewals,
                policy.confirmations, st->solve_ms,
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count(),
                std::max(final_res[0], final_res[1]), st->lambda[1] - st->lambda_next);
    if (!converged) {
        // maps_out and stats are filled: the caller decides whether the block is usable
        scs_set_error("scs_fiedler: residual %.3e above tol %.3e after %d iterations (V = %d, block %d)",
                      std::max(final_res[0], final_res[1]), tol, iter, n, b);
        return SCS_ENOCONV;
    }
    return SCS_OK;
}

// ---------------------------------------------------------------------------
// batched small nodes: tables -> W -> contraction -> S -> Jacobi -> maps
// ---------------------------------------------------------------------------
// Deep levels of the recursion are thousands of nodes of a few to a few dozen taxa
// (reference: scs.py:110-134 at depth; 47 spectral calls with V <= 100 in the reference's
// supertriplets fixture), each with up to thousands of trees, and at any moment the recursion
// knows only a handful of them (the children of the node it has just split).  A launch per
// node with one workgroup walking the node's trees one after the other left the other 255
// CUs idle for milliseconds (round 2: 3.2 us per tree, 16 ms for a 60-taxon node of 5 000
// trees).  Round 3 spreads ONE node over the chip without giving up the tree-ordered sums:
//   k_small_addends  the trees of a node are dealt to workgroups in runs; a workgroup stages
//                    four trees at a time (one per wave: sparse table over the gaps' depths by
//                    wave shuffles) and every thread writes, for its cells (x, y) and each
//                    tree, the ADDEND value(LCA) * weight -- rounded on its own, 0 when the
//                    tree does not join the two taxa -- to addends[tree][cell];
//   k_small_sum      one thread per cell adds its addends in tree order: the reference's sum
//                    (scs.py:644-658; x + 0.0 == x, so the zeros change nothing), bit-identical
//                    to scs_pcg_build's W; the dependent chain is one fp64 add per tree;
//   k_small_finish   one workgroup per node, in LDS: contraction max-reduce over consecutive id
//                    ranges, scipy's degree scaling, the full Jacobi eigen-decomposition and
//                    scikit-learn's embedding conventions.
// K nodes = three launches, one upload, one download.
struct small_batch {
    const int32_t *n_taxa;      // [K] taxa of the node (<= MAXS)
    const int32_t *n_trees;     // [K]
    const int32_t *n_groups;    // [K] vertices after contraction (>= 2)
    const int32_t *tree_ptr;    // [K+1] first tree of node k in tree_w; its tree_off starts at tree_ptr[k] + k
    const int64_t *leaf_ptr;    // [K+1] first leaf slot of node k
    const int32_t *vertex_ptr;  // [K+1] first vertex of node k in maps; its group_start at vertex_ptr[k] + k
    const int32_t *tree_off;    // per node: n_trees + 1 leaf offsets relative to the node's first slot
    const int32_t *leaf_taxon;
    const int32_t *adj_depth;
    const double *adj_val;
    const double *tree_w;
    const int32_t *group_start;  // per node: n_groups + 1 entries, 0 .. n_taxa
    double *maps;                // [total vertices][2]
    double *lambda;              // [K][3]
    double *w_out;               // per node n_groups^2 doubles at w_ptr[k], or null
    const int64_t *w_ptr;
    // work items of k_small_addends: (node, first tree, end tree); of k_small_sum: (node, first cell)
    const int32_t *item_node, *item_t0, *item_t1;
    const int32_t *sum_node, *sum_e0;
    // [K] first addend of node k.  Nodes of at most SMALL_TR_MAX taxa: addends[add_ptr[k] + (x * v0 + y) * ms +
    // tree], ms = trees rounded up to even -- a cell's addends lie in tree order in ONE piece, which the one
    // thread of that cell streams with 16-byte loads (a node of 8 taxa has 28 cells: with the trees outermost
    // its few threads fetched a line per tree).  Larger nodes: addends[add_ptr[k] + tree * v0^2 + x * v0 + y]
    // -- hundreds of cells, neighbouring threads read neighbouring addresses (measured both ways:
    // tools/small_solve_bench.py).  The space reserved is v0^2 * ms either way.
    const int64_t *add_ptr;
    double *addends;
    const int64_t *w0_ptr;   // [K] the node's v0 x v0 uncontracted weights in w0
    double *w0;
};

constexpr int SMALL_Q = MAXS * MAXS / 256;  // cells of a thread at the largest node of the one-wave form
constexpr int SMALL_MAXS = 128;             // largest node of the batched path (round 4: was MAXS)
constexpr int SMALL_TR_MAX = 24;            // up to here a cell's addends are stored contiguously (small_batch::add_ptr)

// A node of 65 .. 128 taxa (round 4; SURVEY.md 8f rank 3 asks for V <= 128): the same three
// launches.  A tree restricted to such a node has up to 128 leaves -- two per lane while a wave
// stages it, the sparse table built level by level in LDS (seven levels, keys (depth << 7 | gap));
// a thread owns the cells of one column y and every other row x < y.
__device__ void small_addends_big(const small_batch &p, int k, int t0, int t1, unsigned (*s_sp)[7][SMALL_MAXS],
                                  double (*s_val)[SMALL_MAXS], int (*s_pos)[SMALL_MAXS]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v0 = p.n_taxa[k];
    const int ncell = v0 * v0;
    const int32_t *toff = p.tree_off + p.tree_ptr[k] + k;
    const int64_t lbase = p.leaf_ptr[k];
    double *out = p.addends + p.add_ptr[k];
    const bool tr = v0 <= SMALL_TR_MAX;  // (never here: the big path starts above 64 taxa; kept for symmetry)
    const int64_t cs = tr ? (p.n_trees[k] + 1) & ~1 : 1;  // stride between two cells of a tree (small_batch::add_ptr)
    const int y = tid & 127, x0 = tid >> 7;
    for (int g = t0; g < t1; g += 4) {
        __syncthreads();  // the cells of the previous four trees are done with the buffers
        const int ts = g + wave;
        if (ts < t1) {
            const int off = toff[ts], n = toff[ts + 1] - off;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = lane + 64 * h;
                int f_dep = 0;
                double f_val = 0.0;
                if (i < n) {
                    f_dep = p.adj_depth[lbase + off + i];
                    f_val = p.adj_val[lbase + off + i];
                }
                // gap i = the LCA of leaves i and i + 1 (adj_* of leaf i), i < n - 1
                s_sp[wave][0][i] = i + 1 < n ? ((unsigned)f_dep << 7) | (unsigned)i : 0xFFFFFFFFu;
                s_val[wave][i] = f_val;
                s_pos[wave][i] = -1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations are performed in order)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = lane + 64 * h;
                if (i < n) s_pos[wave][p.leaf_taxon[lbase + off + i]] = i;
            }
            for (int j = 1; j < 7; ++j) {
                if ((1 << j) >= n) break;  // (uniform) no pair of this tree is 2^j gaps apart
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                unsigned nk[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = lane + 64 * h, i2 = i + (1 << (j - 1));
                    const unsigned a = s_sp[wave][j - 1][i];
                    const unsigned b = i2 < SMALL_MAXS ? s_sp[wave][j - 1][i2] : 0xFFFFFFFFu;
                    nk[h] = b < a ? b : a;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) s_sp[wave][j][lane + 64 * h] = nk[h];
            }
        }
        __syncthreads();
        for (int j = 0; j < 4; ++j) {
            const int t = g + j;
            if (t >= t1) break;
            const double wt = p.tree_w[p.tree_ptr[k] + t];
            double *row = tr ? out + t : out + (int64_t)t * ncell;
            const int py = y < v0 ? s_pos[j][y] : -1;
            for (int x = x0; x < y && y < v0; x += 2) {
                const int px = s_pos[j][x];
                const bool live = px >= 0 && py >= 0;
                const int lo = live ? (px < py ? px : py) : 0, hi = live ? (px < py ? py : px) : 1;
                const int lv = 31 - __clz(hi - lo);
                const unsigned ka = s_sp[j][lv][lo], kb = s_sp[j][lv][hi - (1 << lv)];
                // (leftmost on ties, as a left-to-right sweep finds it)
                const unsigned key = kb < ka ? kb : ka;
                const double mv = s_val[j][key & 127u];
                double add = 0.0;
                // the root (depth 0) separates the two: nothing to add
                if (live && (key >> 7) != 0) {
#pragma clang fp contract(off)
                    add = mv * wt;  // rounded on its own, never fused into the sum
                }
                row[(int64_t)(x * v0 + y) * cs] = add;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_small_addends(small_batch p) {
    // per wave one staged tree: sparse table over the gaps' (depth << 6 | gap) keys, the gaps'
    // values, the position of every taxon (-1: absent)
    __shared__ unsigned s_sp[4][7][SMALL_MAXS];
    __shared__ double s_val[4][SMALL_MAXS];
    __shared__ int s_pos[4][SMALL_MAXS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = p.item_node[blockIdx.x];
    const int t0 = p.item_t0[blockIdx.x], t1 = p.item_t1[blockIdx.x];
    const int v0 = p.n_taxa[k];
    if (v0 > MAXS) {
        small_addends_big(p, k, t0, t1, s_sp, s_val, s_pos);
        return;
    }
    const int ncell = v0 * v0;
    const int32_t *toff = p.tree_off + p.tree_ptr[k] + k;
    const int64_t lbase = p.leaf_ptr[k];
    double *out = p.addends + p.add_ptr[k];
    const bool tr = v0 <= SMALL_TR_MAX;
    const int64_t cs = tr ? (p.n_trees[k] + 1) & ~1 : 1;
    const int nq = (ncell + 255) / 256;
    int cx[SMALL_Q], cy[SMALL_Q];
#pragma unroll
    for (int q = 0; q < SMALL_Q; ++q) {
        const int e = tid + 256 * q;
        cx[q] = e / v0;
        cy[q] = e - cx[q] * v0;
        if (e >= ncell || cx[q] >= cy[q]) cx[q] = -1;  // not a cell of the upper triangle
    }
    for (int g = t0; g < t1; g += 4) {
        __syncthreads();  // the cells of the previous four trees are done with the buffers
        const int ts = g + wave;
        if (ts < t1) {
            const int off = toff[ts], n = toff[ts + 1] - off;
            int f_tax = 0, f_dep = 0;
            double f_val = 0.0;
            if (lane < n) {
                f_tax = p.leaf_taxon[lbase + off + lane];
                f_dep = p.adj_depth[lbase + off + lane];
                f_val = p.adj_val[lbase + off + lane];
            }
            // gap i = the LCA of leaves i and i + 1 (adj_* of leaf i), i < n - 1
            unsigned key = lane + 1 < n ? ((unsigned)f_dep << 6) | (unsigned)lane : 0xFFFFFFFFu;
            s_sp[wave][0][lane] = key;
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                if ((1 << j) >= n) break;  // (uniform) no pair of this tree is 2^j gaps apart
                const unsigned other = __shfl_down(key, 1 << (j - 1), 64);
                if (lane + (1 << (j - 1)) < 64) key = other < key ? other : key;
                s_sp[wave][j][lane] = key;
            }
            s_val[wave][lane] = f_val;
            s_pos[wave][lane] = -1;
            if (lane < n) s_pos[wave][f_tax] = lane;  // (same wave: after the clearing store)
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = g + j;
            if (t >= t1) break;
            const double wt = p.tree_w[p.tree_ptr[k] + t];
            double *row = tr ? out + t : out + (int64_t)t * ncell;
#pragma unroll
            for (int q = 0; q < SMALL_Q; ++q) {
                if (q >= nq) break;  // (uniform) most nodes are tiny
                if (cx[q] < 0) continue;
                const int px = s_pos[j][cx[q]], py = s_pos[j][cy[q]];
                const bool live = px >= 0 && py >= 0;
                const int lo = live ? (px < py ? px : py) : 0, hi = live ? (px < py ? py : px) : 1;
                const int lv = 31 - __clz(hi - lo);
                const unsigned ka = s_sp[j][lv][lo], kb = s_sp[j][lv][hi - (1 << lv)];
                // (leftmost on ties, as a left-to-right sweep finds it)
                const unsigned key = kb < ka ? kb : ka;
                const double mv = s_val[j][key & 63u];
                double add = 0.0;
                // the root (depth 0) separates the two: nothing to add
                if (live && (key >> 6) != 0) {
#pragma clang fp contract(off)
                    add = mv * wt;  // rounded on its own, never fused into the sum
                }
                row[(int64_t)(tid + 256 * q) * cs] = add;
            }
        }
    }
}

// thread = one cell (x < y) of one node: its addends in tree order (reference: scs.py:655-657)
__global__ __launch_bounds__(256) void k_small_sum(small_batch p) {
    const int k = p.sum_node[blockIdx.x];
    const int e = p.sum_e0[blockIdx.x] + threadIdx.x;
    const int v0 = p.n_taxa[k], m = p.n_trees[k];
    const int ncell = v0 * v0;
    if (e >= ncell) return;
    const int x = e / v0, y = e - x * v0;
    double *w0 = p.w0 + p.w0_ptr[k];
    if (x == y) w0[e] = 0.0;
    if (x >= y) return;
    double acc = 0.0;
    int t = 0;
    // (the adds are one chain in tree order -- the reference's order, scs.py:656 -- but the LOADS need not wait
    // for it.  With 8 loads in flight the kernel took 190 us for 5 000 trees whatever the node's size, 625 round
    // trips, and a recursion runs it tens of thousands of times; tools/small_solve_bench.py)
    if (v0 <= SMALL_TR_MAX) {
        const int64_t ms = (m + 1) & ~1;
        const double *in = p.addends + p.add_ptr[k] + (int64_t)e * ms;  // 16-byte aligned: add_ptr and ms are even
        // two batches of 64 addends in flight: the next one's loads are issued before this one's adds
        if (t + 64 <= m) {
            double2 a[32], b[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j] = *(const double2 *)(in + t + 2 * j);
            for (; t + 128 <= m; t += 64) {
#pragma unroll
                for (int j = 0; j < 32; ++j) b[j] = *(const double2 *)(in + t + 64 + 2 * j);
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    acc = acc + a[j].x;
                    acc = acc + a[j].y;
                }
#pragma unroll
                for (int j = 0; j < 32; ++j) a[j] = b[j];
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                acc = acc + a[j].x;
                acc = acc + a[j].y;
            }
            t += 64;
        }
        for (; t + 8 <= m; t += 8) {
            double2 a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *(const double2 *)(in + t + 2 * j);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = acc + a[j].x;
                acc = acc + a[j].y;
            }
        }
        for (; t < m; ++t) acc = acc + in[t];
    } else {
        const double *in = p.addends + p.add_ptr[k] + e;
        for (; t + 32 <= m; t += 32) {
            double a[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) a[j] = in[(int64_t)(t + j) * ncell];
#pragma unroll
            for (int j = 0; j < 32; ++j) acc = acc + a[j];
        }
        for (; t < m; ++t) acc = acc + in[(int64_t)t * ncell];
    }
    w0[e] = acc;
    w0[y * v0 + x] = acc;
}

// NT = 256: the nodes of up to 24 vertices (the Jacobi variants written for 256 threads).  NT = 1 024 (round 5, late):
// the nodes of 25 .. 64 vertices -- everything up to the eigen-solve is done by the first 256 threads as it
// always was, the Jacobi sweeps spread their independent work items over all 1 024 (jacobi_eig_generic): the same
// bits, a solve of 64 vertices 1.5 -> 0.7 ms.  The host launches the instance(s) a batch needs.
template <int NT>
__global__ __launch_bounds__(NT) void k_small_finish(small_batch p) {
    __shared__ jacobi_lds s;
    __shared__ double s_dd[MAXS];
    __shared__ int s_gs[MAXS + 1];
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    const int v = p.n_groups[k], v0 = p.n_taxa[k];
    if (v0 > MAXS) return;  // k_small_finish_big's
    if ((v > 24) != (NT > 256)) return;  // the other instance's
    const bool first = tid < 256;  // the threads of the 256-thread form: every strided loop below is theirs
    double(*w0)[SLD] = s.e;  // the uncontracted weights live where Jacobi later keeps its vectors
    if (first && tid <= v) s_gs[tid] = p.group_start[p.vertex_ptr[k] + k + tid];
    {
        const double *src = p.w0 + p.w0_ptr[k];
        for (int e = tid; e < v0 * v0 && first; e += 256) w0[e / v0][e % v0] = src[e];
    }
    __syncthreads();
    // ---- contraction: vertex g = taxa [gs[g], gs[g+1]); weight = max over member pairs
    // (reference: scs.py:336-387), diagonal 0
    for (int e = tid; e < v * v && first; e += 256) {
        const int g = e / v, h = e - g * v;
        double best = 0.0;
        if (g != h) {
            best = w0[s_gs[g]][s_gs[h]];
            for (int r = s_gs[g]; r < s_gs[g + 1]; ++r)
                for (int c = s_gs[h]; c < s_gs[h + 1]; ++c) best = w0[r][c] > best ? w0[r][c] : best;
        }
        s.a[g][h] = best;
    }
    __syncthreads();
    if (p.w_out)
        for (int e = tid; e < v * v && first; e += 256) p.w_out[p.w_ptr[k] + e] = s.a[e / v][e % v];
    // ---- degrees as scipy takes them (column sums, rows in order; isolated -> 1)
    if (tid < v) {
        double d = 0.0;
        for (int i = 0; i < v; ++i) d = d + s.a[i][tid];
        s_dd[tid] = d == 0.0 ? 1.0 : sqrt(d);
    }
    __syncthreads();
    // S = (W / dd) / dd^T: two successive divisions (scipy/sparse/csgraph/_laplacian.py:552-557),
    // then the symmetric part (the two orders of division may differ in the last bit)
    double keep[(MAXS * MAXS + 255) / 256];
    {
        int q = 0;
        for (int e = tid; e < v * v && first; e += 256, ++q) {
            const int g = e / v, h = e - g * v;
            const double x = (s.a[g][h] / s_dd[h]) / s_dd[g];
            const double y = (s.a[h][g] / s_dd[g]) / s_dd[h];
            keep[q] = g == h ? 0.0 : 0.5 * (x + y);
        }
        __syncthreads();
        q = 0;
        for (int e = tid; e < v * v && first; e += 256, ++q) s.a[e / v][e % v] = keep[q];
    }
    __syncthreads();
    if (NT > 256) jacobi_eig_generic<NT>(s, v);
    else jacobi_eig(s, v);
    // ---- embedding: unit eigenvectors / dd, largest |entry| of each column positive, column 0
    // <-> the largest eigenvalue (sklearn/manifold/_spectral_embedding.py:373-376, 463)
    if (tid < 2) {
        const int col = s.perm[tid];
        int arg = 0;
        double best = -1.0;
        for (int i = 0; i < v; ++i) {
            const double x = fabs(s.e[i][col] / s_dd[i]);
            if (x > best) {
                best = x;
                arg = i;
            }
        }
        const double sg = s.e[arg][col] < 0.0 ? -1.0 : 1.0;
        double *out = p.maps + (int64_t)p.vertex_ptr[k] * 2 + tid;
        for (int i = 0; i < v; ++i) out[2 * i] = sg * (s.e[i][col] / s_dd[i]);
    }
    if (tid < 3) p.lambda[k * 3 + tid] = tid < v ? s.w[tid] : 0.0;
}

// The nodes of 65 .. 128 taxa of a batch, one workgroup each: contraction straight from the
// uncontracted weights in device memory into the one LDS array the one-sided Jacobi works in
// (dense path of scs_fiedler: (V + 1) x V doubles, 132 KB at 128), scipy's degree scaling in
// place (the contracted matrix is symmetric bit for bit, so every cell's two orders of division
// need only the cell itself), A = S + I, the sweeps, scikit-learn's embedding conventions.  A
// batch's workgroups of this kind run side by side on as many CUs: what the recursion's walk
// used to solve one node after the other with ~35 latency-bound LOBPCG iterations each.
__global__ __launch_bounds__(256) void k_small_finish_big(small_batch p) {
    extern __shared__ double dyn_lds[];
    double(*a)[DENSE2_LD] = (double(*)[DENSE2_LD])dyn_lds;
    __shared__ double s_dd[SMALL_MAXS], s_norm[SMALL_MAXS];
    __shared__ int s_gs[SMALL_MAXS + 1];
    __shared__ int s_rot, s_top[3];
    const int tid = threadIdx.x;
    const int k = blockIdx.x;
    const int v = p.n_groups[k], v0 = p.n_taxa[k];
    if (v0 <= MAXS) return;  // k_small_finish's
    const int m = v + (v & 1);
    if (tid <= v) s_gs[tid] = p.group_start[p.vertex_ptr[k] + k + tid];
    __syncthreads();
    // ---- contraction: vertex g = taxa [gs[g], gs[g+1]); weight = max over member pairs
    // (reference: scs.py:336-387), diagonal 0; the padding row / column of an odd V is zero
    const double *w0 = p.w0 + p.w0_ptr[k];
    for (int e = tid; e < m * m; e += 256) {
        const int g = e / m, h = e - g * m;
        double best = 0.0;
        if (g != h && g < v && h < v) {
            best = w0[s_gs[g] * v0 + s_gs[h]];
            for (int r = s_gs[g]; r < s_gs[g + 1]; ++r)
                for (int c = s_gs[h]; c < s_gs[h + 1]; ++c) {
                    const double x = w0[r * v0 + c];
                    best = x > best ? x : best;
                }
        }
        a[g][h] = best;
    }
    __syncthreads();
    if (p.w_out)
        for (int e = tid; e < v * v; e += 256) p.w_out[p.w_ptr[k] + e] = a[e / v][e % v];
    // ---- degrees as scipy takes them (column sums, rows in order; isolated -> 1)
    if (tid < v) {
        double d = 0.0;
        for (int i = 0; i < v; ++i) d = d + a[i][tid];
        s_dd[tid] = d == 0.0 ? 1.0 : sqrt(d);
    }
    __syncthreads();
    // S = (W / dd) / dd^T: two successive divisions (scipy/sparse/csgraph/_laplacian.py:552-557),
    // then the symmetric part (the two orders of division may differ in the last bit); + I
    for (int e = tid; e < v * v; e += 256) {
        const int g = e / v, h = e - g * v;
        const double w = a[g][h];  // == a[h][g]
        const double x = (w / s_dd[h]) / s_dd[g];
        const double y = (w / s_dd[g]) / s_dd[h];
        a[g][h] = g == h ? 1.0 : 0.5 * (x + y);
    }
    __syncthreads();
    const bool ok = onesided_sweeps(a, m, &s_rot);
    // eigenvalues of S: column norms - 1 (the padding column of an odd V has norm 0: last)
    if (tid < v) {
        double s2 = 0.0;
        for (int r = 0; r < v; ++r) s2 = fma(a[r][tid], a[r][tid], s2);
        s_norm[tid] = sqrt(s2);
    }
    __syncthreads();
    if (tid < v) {
        const double mine = s_norm[tid];
        int rank = 0;
        for (int j = 0; j < v; ++j) {
            const double o = s_norm[j];
            if (o > mine || (o == mine && j < tid)) ++rank;
        }
        if (rank < 3) s_top[rank] = tid;
    }
    __syncthreads();
    // ---- embedding: unit eigenvectors / dd, largest |entry| of each column positive, column 0
    // <-> the largest eigenvalue (sklearn/manifold/_spectral_embedding.py:373-376, 463)
    if (tid < 2) {
        const int col = s_top[tid];
        const double nrm = s_norm[col];
        int arg = 0;
        double best = -1.0;
        for (int i = 0; i < v; ++i) {
            const double x = fabs((a[i][col] / nrm) / s_dd[i]);
            if (x > best) {
                best = x;
                arg = i;
            }
        }
        const double sg = a[arg][col] < 0.0 ? -1.0 : 1.0;
        double *out = p.maps + (int64_t)p.vertex_ptr[k] * 2 + tid;
        for (int i = 0; i < v; ++i) out[2 * i] = sg * ((a[i][col] / nrm) / s_dd[i]);
    }
    // (sweeps that did not converge: NaN eigenvalues -- the host refuses the node)
    if (tid < 3) p.lambda[k * 3 + tid] = !ok ? __longlong_as_double(0x7FF8000000000000ll)
                                             : (tid < v ? s_norm[s_top[tid]] - 1.0 : 0.0);
}

// leaf arrays of ONE node straight from a forest's device tables into the batch's device block
// (taxa renumbered through `relabel` when the node numbers them differently -- present taxa only,
// contraction groups made consecutive), offsets narrowed to 32 bits
__global__ void k_small_pack(const int64_t *__restrict__ tree_off64, const int32_t *__restrict__ leaf_taxon,
                             const int32_t *__restrict__ adj_depth, const double *__restrict__ adj_val,
                             const double *__restrict__ tree_w, const int32_t *__restrict__ relabel,
                             int32_t n_trees, int64_t n_leaves, int32_t *__restrict__ to, int32_t *__restrict__ lt,
                             int32_t *__restrict__ ad, double *__restrict__ av, double *__restrict__ tw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_leaves) {
        const int32_t x = leaf_taxon[i];
        lt[i] = relabel ? relabel[x] : x;
        ad[i] = adj_depth[i];
        av[i] = adj_val[i];
    }
    if (i <= n_trees) to[i] = (int32_t)tree_off64[i];
    if (i < n_trees) tw[i] = tree_w[i];
}

// the same for the K nodes of a level (scs_small_solve_begin_level): node k owns the trees [t_begin[k],
// t_begin[k] + n_trees[k]) of the level forest and the taxon ids [u_base[k], u_base[k] + u_size[k]) of its
// universe; relabel (all nodes' maps, concatenated: rl_ptr[k] is node k's first entry) sends an id of that
// range to the node's own numbering
struct level_pack {
    const int32_t *t_begin, *u_base, *rl_ptr;  // device [K]
    const int32_t *relabel;                    // device, concatenated
    const int32_t *tree_ptr;                   // device [K + 1] first tree of node k in the batch
    const int64_t *leaf_ptr;                   // device [K + 1] first leaf slot of node k in the batch
};

__global__ void k_small_pack_level(const int64_t *__restrict__ tree_off64, const int32_t *__restrict__ leaf_taxon,
                                   const int32_t *__restrict__ adj_depth, const double *__restrict__ adj_val,
                                   const double *__restrict__ tree_w, level_pack lp, int32_t n_nodes, int64_t n_leaves,
                                   int32_t n_trees, int32_t *__restrict__ to, int32_t *__restrict__ lt,
                                   int32_t *__restrict__ ad, double *__restrict__ av, double *__restrict__ tw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_leaves) {
        int32_t lo = 0, hi = n_nodes - 1;  // last k with leaf_ptr[k] <= i
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (lp.leaf_ptr[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        const int64_t src = tree_off64[lp.t_begin[lo]] + (i - lp.leaf_ptr[lo]);
        lt[i] = lp.relabel[lp.rl_ptr[lo] + (leaf_taxon[src] - lp.u_base[lo])];
        ad[i] = adj_depth[src];
        av[i] = adj_val[src];
    }
    if (i < n_trees + n_nodes) {
        // node k's offsets sit at tree_ptr[k] + k .. tree_ptr[k + 1] + k (its trees + 1 entries)
        int32_t lo = 0, hi = n_nodes - 1;  // last k with tree_ptr[k] + k <= i
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (lp.tree_ptr[mid] + mid <= i) lo = mid;
            else hi = mid - 1;
        }
        const int32_t j = (int32_t)i - (lp.tree_ptr[lo] + lo);  // 0 .. trees of the node
        const int32_t t0 = lp.t_begin[lo];
        to[i] = (int32_t)(tree_off64[t0 + j] - tree_off64[t0]);
        if (j < lp.tree_ptr[lo + 1] - lp.tree_ptr[lo]) tw[lp.tree_ptr[lo] + j] = tree_w[t0 + j];
    }
}

struct level_src {
    const scs_forest *forest;
    const int32_t *t_begin, *u_base, *u_size;  // host [K]
    const int64_t *n_leaves;                   // host [K]
    const int32_t *relabel;                    // host, concatenated (sum of u_size entries)
};

static int small_solve_begin_impl(scs_ctx *ctx, int32_t n_nodes, const int32_t *n_taxa,
                                  const int32_t *n_trees, const int32_t *n_groups,
                                  const int32_t *tree_off, const int32_t *leaf_taxon,
                                  const int32_t *adj_depth, const double *adj_val,
                                  const double *tree_w, const int32_t *group_start, int32_t want_w,
                                  int32_t *ticket_out, const scs_forest *src, const int32_t *relabel,
                                  const level_src *lvl = nullptr);

// The K small nodes of one level of the recursion straight from the level forest's resident tables
// (scs_forest_split_level): nothing but the renumbering and the group boundaries travels.
extern "C" int scs_small_solve_begin_level(scs_ctx *ctx, const scs_forest *forest, int32_t n_nodes,
                                           const int32_t *t_begin, const int32_t *n_trees, const int64_t *n_leaves,
                                           const int32_t *u_base, const int32_t *u_size, const int32_t *relabel,
                                           const int32_t *n_taxa, const int32_t *n_groups,
                                           const int32_t *group_start, int32_t want_w, int32_t *ticket_out) {
    SCS_REQUIRE(ctx && forest && t_begin && n_trees && n_leaves && u_base && u_size && relabel && n_taxa && n_groups &&
                    group_start && ticket_out,
                "scs_small_solve_begin_level: null argument");
    SCS_REQUIRE(forest->has_tables, "scs_small_solve_begin_level: the forest carries no tables");
    for (int32_t k = 0; k < n_nodes; ++k) {
        SCS_REQUIRE(t_begin[k] >= 0 && n_trees[k] >= 1 && (int64_t)t_begin[k] + n_trees[k] <= forest->n_trees,
                    "scs_small_solve_begin_level: node %d: bad tree range", k);
        SCS_REQUIRE(u_base[k] >= 0 && u_size[k] >= 1 && (int64_t)u_base[k] + u_size[k] <= forest->n_taxa,
                    "scs_small_solve_begin_level: node %d: bad taxon range", k);
        SCS_REQUIRE(n_leaves[k] >= 2 * (int64_t)n_trees[k] && n_leaves[k] <= (int64_t)n_trees[k] * n_taxa[k],
                    "scs_small_solve_begin_level: node %d: bad leaf count", k);
    }
    level_src lvl{forest, t_begin, u_base, u_size, n_leaves, relabel};
    return small_solve_begin_impl(ctx, n_nodes, n_taxa, n_trees, n_groups, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  group_start, want_w, ticket_out, nullptr, nullptr, &lvl);
}

extern "C" int scs_small_solve_begin(scs_ctx *ctx, int32_t n_nodes, const int32_t *n_taxa,
                                     const int32_t *n_trees, const int32_t *n_groups,
                                     const int32_t *tree_off, const int32_t *leaf_taxon,
                                     const int32_t *adj_depth, const double *adj_val,
                                     const double *tree_w, const int32_t *group_start, int32_t want_w,
                                     int32_t *ticket_out) {
    SCS_REQUIRE(ctx && n_taxa && n_trees && n_groups && tree_off && leaf_taxon && adj_depth &&
                    adj_val && tree_w && group_start && ticket_out,
                "scs_small_solve: null argument");
    return small_solve_begin_impl(ctx, n_nodes, n_taxa, n_trees, n_groups, tree_off, leaf_taxon, adj_depth, adj_val,
                                  tree_w, group_start, want_w, ticket_out, nullptr, nullptr);
}

// ONE node whose tables are resident (a child of scs_forest_split): nothing but the group boundaries
// and the renumbering travels; the leaf arrays are packed on the device (k_small_pack)
extern "C" int scs_small_solve_begin_forest(scs_ctx *ctx, const scs_forest *forest, const int32_t *relabel,
                                            int32_t n_taxa, int32_t n_groups, const int32_t *group_start,
                                            int32_t want_w, int32_t *ticket_out) {
    SCS_REQUIRE(ctx && forest && group_start && ticket_out, "scs_small_solve_begin_forest: null argument");
    SCS_REQUIRE(forest->has_tables, "scs_small_solve_begin_forest: the forest carries no tables (not a child of scs_forest_split)");
    SCS_REQUIRE(forest->n_leaves <= INT32_MAX, "scs_small_solve_begin_forest: too many leaves");
    const int32_t n_trees = forest->n_trees;
    return small_solve_begin_impl(ctx, 1, &n_taxa, &n_trees, &n_groups, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  group_start, want_w, ticket_out, forest, relabel);
}

static int small_solve_begin_impl(scs_ctx *ctx, int32_t n_nodes, const int32_t *n_taxa,
                                  const int32_t *n_trees, const int32_t *n_groups,
                                  const int32_t *tree_off, const int32_t *leaf_taxon,
                                  const int32_t *adj_depth, const double *adj_val,
                                  const double *tree_w, const int32_t *group_start, int32_t want_w,
                                  int32_t *ticket_out, const scs_forest *src, const int32_t *relabel,
                                  const level_src *lvl) {
    const bool w_out = want_w != 0;
    SCS_REQUIRE(n_nodes >= 1, "scs_small_solve: need at least one node");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    // (a stream of its own: scs_internal.h)
    const bool own_stream = true;
    if (own_stream && !ctx->small_stream)
        SCS_HIP_CHECK(hipStreamCreateWithFlags(&ctx->small_stream, hipStreamNonBlocking));
    hipStream_t s = own_stream ? ctx->small_stream : ctx->stream;
    const int K = n_nodes;
    // ---- layout of the one staging block: [pointers | tree_off | group_start | leaf arrays |
    // tree_w], 8-byte aligned pieces
    std::vector<int32_t> tree_ptr(K + 1, 0), vertex_ptr(K + 1, 0);
    std::vector<int64_t> leaf_ptr(K + 1, 0), w_ptr(K + 1, 0);
    int64_t toff_at = 0;
    bool any_big = false;  // nodes of more than MAXS taxa: k_small_finish_big
    for (int k = 0; k < K; ++k) {
        SCS_REQUIRE(n_taxa[k] >= 2 && n_taxa[k] <= SMALL_MAXS, "scs_small_solve: node %d has %d taxa (2..%d)",
                    k, n_taxa[k], SMALL_MAXS);
        any_big = any_big || n_taxa[k] > MAXS;
        SCS_REQUIRE(n_groups[k] >= 2 && n_groups[k] <= n_taxa[k], "scs_small_solve: node %d: bad group count %d",
                    k, n_groups[k]);
        SCS_REQUIRE(n_trees[k] >= 1, "scs_small_solve: node %d has no tree", k);
        const int32_t *to = (src || lvl) ? nullptr : tree_off + toff_at;
        if (!src && !lvl) {
            SCS_REQUIRE(to[0] == 0, "scs_small_solve: node %d: tree_off must start at 0", k);
            for (int t = 0; t < n_trees[k]; ++t)
                SCS_REQUIRE(to[t + 1] >= to[t] && to[t + 1] - to[t] <= n_taxa[k],
                            "scs_small_solve: node %d tree %d: bad leaf range", k, t);
        }
        const int32_t *gs = group_start + vertex_ptr[k] + k;
        SCS_REQUIRE(gs[0] == 0 && gs[n_groups[k]] == n_taxa[k], "scs_small_solve: node %d: group_start must span the taxa", k);
        for (int g = 0; g < n_groups[k]; ++g)
            SCS_REQUIRE(gs[g] < gs[g + 1], "scs_small_solve: node %d: empty group %d", k, g);
        tree_ptr[k + 1] = tree_ptr[k] + n_trees[k];
        leaf_ptr[k + 1] = leaf_ptr[k] + (lvl ? lvl->n_leaves[k] : src ? src->n_leaves : (int64_t)to[n_trees[k]]);
        vertex_ptr[k + 1] = vertex_ptr[k] + n_groups[k];
        w_ptr[k + 1] = w_ptr[k] + (int64_t)n_groups[k] * n_groups[k];
        toff_at += n_trees[k] + 1;
    }
    const int64_t n_toff = toff_at, n_gs = vertex_ptr[K] + K, n_leaf = leaf_ptr[K], n_tree = tree_ptr[K];
    // ---- work items: runs of trees for k_small_addends (enough runs to fill the chip, at
    // least 16 trees each, whole groups of four), 256-cell pieces for k_small_sum
    int per_item = (int)std::max<int64_t>(16, (n_tree + 1023) / 1024);
    per_item = (per_item + 3) / 4 * 4;
    std::vector<int32_t> item_node, item_t0, item_t1, sum_node, sum_e0;
    std::vector<int64_t> add_ptr(K + 1, 0), w0_ptr(K + 1, 0);
    for (int k = 0; k < K; ++k) {
        const int64_t ncell = (int64_t)n_taxa[k] * n_taxa[k];
        add_ptr[k + 1] = add_ptr[k] + ncell * (((int64_t)n_trees[k] + 1) & ~(int64_t)1);
        w0_ptr[k + 1] = w0_ptr[k] + ncell;
        for (int t = 0; t < n_trees[k]; t += per_item) {
            item_node.push_back(k);
            item_t0.push_back(t);
            item_t1.push_back(std::min(n_trees[k], t + per_item));
        }
        for (int e = 0; e < ncell; e += 256) {
            sum_node.push_back(k);
            sum_e0.push_back(e);
        }
    }
    const size_t n_items = item_node.size(), n_sums = sum_node.size();
    SCS_REQUIRE((uint64_t)add_ptr[K] * 8 <= ((uint64_t)48 << 30),
                "scs_small_solve: batch too large (%lld addends): pass fewer nodes per call",
                (long long)add_ptr[K]);
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    size_t at = 0;
    const size_t o_nt = at; at += up8((size_t)K * 4);
    const size_t o_nm = at; at += up8((size_t)K * 4);
    const size_t o_ng = at; at += up8((size_t)K * 4);
    const size_t o_tp = at; at += up8((size_t)(K + 1) * 4);
    const size_t o_vp = at; at += up8((size_t)(K + 1) * 4);
    const size_t o_lp = at; at += (size_t)(K + 1) * 8;
    const size_t o_wp = at; at += (size_t)(K + 1) * 8;
    const size_t o_ap = at; at += (size_t)(K + 1) * 8;
    const size_t o_w0p = at; at += (size_t)(K + 1) * 8;
    const size_t o_in = at; at += up8(n_items * 4);
    const size_t o_i0 = at; at += up8(n_items * 4);
    const size_t o_i1 = at; at += up8(n_items * 4);
    const size_t o_sn = at; at += up8(n_sums * 4);
    const size_t o_s0 = at; at += up8(n_sums * 4);
    const size_t o_to = at; at += up8((size_t)n_toff * 4);
    const size_t o_gs = at; at += up8((size_t)n_gs * 4);
    const size_t o_lt = at; at += up8((size_t)n_leaf * 4);
    const size_t o_ad = at; at += up8((size_t)n_leaf * 4);
    const size_t o_av = at; at += (size_t)n_leaf * 8;
    const size_t o_tw = at; at += (size_t)n_tree * 8;
    // (a resident node: the leaf arrays above are filled on the device; only the header travels --
    // up to o_to -- and the renumbering behind it)
    const size_t o_rl = at; if (src && relabel) at += up8((size_t)src->n_taxa * 4);
    // a level: t_begin | u_base | rl_ptr [K] each, then the concatenated renumbering
    std::vector<int32_t> rl_ptr(lvl ? K + 1 : 0, 0);
    for (int k = 0; lvl && k < K; ++k) rl_ptr[k + 1] = rl_ptr[k] + lvl->u_size[k];
    const size_t o_ltb = at; if (lvl) at += up8((size_t)K * 4);
    const size_t o_lub = at; if (lvl) at += up8((size_t)K * 4);
    const size_t o_lrp = at; if (lvl) at += up8((size_t)K * 4);
    const size_t o_lrl = at; if (lvl) at += up8((size_t)rl_ptr[K] * 4);
    const size_t in_bytes = at;
    const size_t o_maps = at; at += (size_t)vertex_ptr[K] * 16;
    const size_t o_lam = at; at += (size_t)K * 24;
    const size_t o_w = at; if (w_out) at += (size_t)w_ptr[K] * 8;
    const size_t total = at;
    // a free slot that is large enough (the smallest such), else the largest free one (grown), else
    // a new one
    int pick = -1;
    for (size_t i = 0; i < ctx->small_slots.size(); ++i) {
        const auto &c = ctx->small_slots[i];
        if (c.busy) continue;
        if (pick < 0) {
            pick = (int)i;
            continue;
        }
        const auto &p = ctx->small_slots[pick];
        const bool c_fits = c.cap >= total, p_fits = p.cap >= total;
        if ((c_fits && (!p_fits || c.cap < p.cap)) || (!c_fits && !p_fits && c.cap > p.cap)) pick = (int)i;
    }
    if (pick < 0) {
        ctx->small_slots.emplace_back();
        pick = (int)ctx->small_slots.size() - 1;
    }
    scs_ctx::small_slot &slot = ctx->small_slots[pick];
    if (slot.cap < total) {
        if (slot.dev) scs_dev_free(slot.dev);
        if (slot.host) hipHostFree(slot.host);
        slot.dev = nullptr;
        slot.host = nullptr;
        slot.cap = 0;
        const size_t cap = std::max<size_t>(total * 2, (size_t)1 << 20);
        SCS_HIP_CHECK(scs_dev_malloc(ctx, (void **)&slot.dev, cap));
        SCS_HIP_CHECK(hipHostMalloc((void **)&slot.host, cap, hipHostMallocDefault));
        slot.cap = cap;
    }
    if (!slot.done) SCS_HIP_CHECK(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    // device scratch: the addends (trees x cells per node) and the uncontracted weights
    t_ctx = ctx;
    const size_t add_bytes = ((size_t)std::max<int64_t>(add_ptr[K], 1) * 8 + 255) / 256 * 256;
    const size_t w0_bytes = (size_t)std::max<int64_t>(w0_ptr[K], 1) * 8;
    if (slot.scratch_cap < add_bytes + w0_bytes) {
        if (slot.scratch) scs_dev_free(slot.scratch);
        slot.scratch = nullptr;
        slot.scratch_cap = 0;
        const size_t cap = std::max<size_t>((add_bytes + w0_bytes) * 3 / 2, (size_t)1 << 20);
        SCS_HIP_CHECK(scs_dev_malloc(ctx, (void **)&slot.scratch, cap));
        slot.scratch_cap = cap;
    }
    unsigned char *h = slot.host, *d = slot.dev;
    memcpy(h + o_nt, n_taxa, (size_t)K * 4);
    memcpy(h + o_nm, n_trees, (size_t)K * 4);
    memcpy(h + o_ng, n_groups, (size_t)K * 4);
    memcpy(h + o_tp, tree_ptr.data(), (size_t)(K + 1) * 4);
    memcpy(h + o_vp, vertex_ptr.data(), (size_t)(K + 1) * 4);
    memcpy(h + o_lp, leaf_ptr.data(), (size_t)(K + 1) * 8);
    memcpy(h + o_wp, w_ptr.data(), (size_t)(K + 1) * 8);
    memcpy(h + o_ap, add_ptr.data(), (size_t)(K + 1) * 8);
    memcpy(h + o_w0p, w0_ptr.data(), (size_t)(K + 1) * 8);
    memcpy(h + o_in, item_node.data(), n_items * 4);
    memcpy(h + o_i0, item_t0.data(), n_items * 4);
    memcpy(h + o_i1, item_t1.data(), n_items * 4);
    memcpy(h + o_sn, sum_node.data(), n_sums * 4);
    memcpy(h + o_s0, sum_e0.data(), n_sums * 4);
    memcpy(h + o_gs, group_start, (size_t)n_gs * 4);
    if (lvl) {
        // header and group boundaries, the level's per-node ranges and renumbering; the leaf arrays are
        // packed on the device from the level forest's tables
        memcpy(h + o_ltb, lvl->t_begin, (size_t)K * 4);
        memcpy(h + o_lub, lvl->u_base, (size_t)K * 4);
        memcpy(h + o_lrp, rl_ptr.data(), (size_t)K * 4);
        memcpy(h + o_lrl, lvl->relabel, (size_t)rl_ptr[K] * 4);
        SCS_HIP_CHECK(hipMemcpyAsync(d, h, o_to, hipMemcpyHostToDevice, s));
        SCS_HIP_CHECK(hipMemcpyAsync(d + o_gs, h + o_gs, up8((size_t)n_gs * 4), hipMemcpyHostToDevice, s));
        SCS_HIP_CHECK(hipMemcpyAsync(d + o_ltb, h + o_ltb, in_bytes - o_ltb, hipMemcpyHostToDevice, s));
        level_pack lp;
        lp.t_begin = (const int32_t *)(d + o_ltb);
        lp.u_base = (const int32_t *)(d + o_lub);
        lp.rl_ptr = (const int32_t *)(d + o_lrp);
        lp.relabel = (const int32_t *)(d + o_lrl);
        lp.tree_ptr = (const int32_t *)(d + o_tp);
        lp.leaf_ptr = (const int64_t *)(d + o_lp);
        const scs_forest *lf = lvl->forest;
        const int64_t work = std::max<int64_t>(n_leaf, n_tree + K);
        k_small_pack_level<<<(unsigned)((work + 255) / 256), 256, 0, s>>>(
            lf->tree_off, lf->leaf_taxon, lf->adj_depth, lf->adj_val, lf->weights, lp, K, n_leaf, (int32_t)n_tree,
            (int32_t *)(d + o_to), (int32_t *)(d + o_lt), (int32_t *)(d + o_ad), (double *)(d + o_av),
            (double *)(d + o_tw));
    } else if (!src) {
        memcpy(h + o_to, tree_off, (size_t)n_toff * 4);
        memcpy(h + o_lt, leaf_taxon, (size_t)n_leaf * 4);
        memcpy(h + o_ad, adj_depth, (size_t)n_leaf * 4);
        memcpy(h + o_av, adj_val, (size_t)n_leaf * 8);
        memcpy(h + o_tw, tree_w, (size_t)n_tree * 8);
        SCS_HIP_CHECK(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    } else {
        // header and group boundaries (two small pieces around the leaf arrays), the renumbering
        SCS_HIP_CHECK(hipMemcpyAsync(d, h, o_to, hipMemcpyHostToDevice, s));
        SCS_HIP_CHECK(hipMemcpyAsync(d + o_gs, h + o_gs, up8((size_t)n_gs * 4), hipMemcpyHostToDevice, s));
        const int32_t *d_rl = nullptr;
        if (relabel) {
            memcpy(h + o_rl, relabel, (size_t)src->n_taxa * 4);
            SCS_HIP_CHECK(hipMemcpyAsync(d + o_rl, h + o_rl, (size_t)src->n_taxa * 4, hipMemcpyHostToDevice, s));
            d_rl = (const int32_t *)(d + o_rl);
        }
        const int64_t work = std::max<int64_t>(n_leaf, n_tree + 1);
        k_small_pack<<<(unsigned)((work + 255) / 256), 256, 0, s>>>(
            src->tree_off, src->leaf_taxon, src->adj_depth, src->adj_val, src->weights, d_rl, (int32_t)n_tree, n_leaf,
            (int32_t *)(d + o_to), (int32_t *)(d + o_lt), (int32_t *)(d + o_ad), (double *)(d + o_av),
            (double *)(d + o_tw));
    }
    small_batch sb;
    sb.n_taxa = (const int32_t *)(d + o_nt);
    sb.n_trees = (const int32_t *)(d + o_nm);
    sb.n_groups = (const int32_t *)(d + o_ng);
    sb.tree_ptr = (const int32_t *)(d + o_tp);
    sb.vertex_ptr = (const int32_t *)(d + o_vp);
    sb.leaf_ptr = (const int64_t *)(d + o_lp);
    sb.w_ptr = (const int64_t *)(d + o_wp);
    sb.add_ptr = (const int64_t *)(d + o_ap);
    sb.w0_ptr = (const int64_t *)(d + o_w0p);
    sb.item_node = (const int32_t *)(d + o_in);
    sb.item_t0 = (const int32_t *)(d + o_i0);
    sb.item_t1 = (const int32_t *)(d + o_i1);
    sb.sum_node = (const int32_t *)(d + o_sn);
    sb.sum_e0 = (const int32_t *)(d + o_s0);
    sb.tree_off = (const int32_t *)(d + o_to);
    sb.group_start = (const int32_t *)(d + o_gs);
    sb.leaf_taxon = (const int32_t *)(d + o_lt);
    sb.adj_depth = (const int32_t *)(d + o_ad);
    sb.adj_val = (const double *)(d + o_av);
    sb.tree_w = (const double *)(d + o_tw);
    sb.maps = (double *)(d + o_maps);
    sb.lambda = (double *)(d + o_lam);
    sb.w_out = w_out ? (double *)(d + o_w) : nullptr;
    sb.addends = (double *)slot.scratch;
    sb.w0 = (double *)(slot.scratch + add_bytes);
    k_small_addends<<<(unsigned)n_items, 256, 0, s>>>(sb);
    k_small_sum<<<(unsigned)n_sums, 256, 0, s>>>(sb);
    {
        bool any_small = false, any_mid = false;
        for (int k = 0; k < K; ++k) {
            if (n_taxa[k] > MAXS) continue;
            if (n_groups[k] > 24) any_mid = true;
            else any_small = true;
        }
        if (any_small) k_small_finish<256><<<K, 256, 0, s>>>(sb);
        if (any_mid) k_small_finish<1024><<<K, 1024, 0, s>>>(sb);
    }
    if (any_big) {
        // (every node gets a workgroup of either kind; the one that is not its own returns at once)
        SCS_HIP_CHECK(hipFuncSetAttribute((const void *)k_small_finish_big, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          DENSE2_MAX * DENSE2_LD * (int)sizeof(double)));
        k_small_finish_big<<<K, 256, (size_t)DENSE2_MAX * DENSE2_LD * sizeof(double), s>>>(sb);
    }
    SCS_HIP_CHECK(hipGetLastError());
    SCS_HIP_CHECK(hipMemcpyAsync(h + o_maps, d + o_maps, total - o_maps, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipEventRecord(slot.done, s));
    slot.busy = true;
    slot.o_maps = o_maps;
    slot.maps_bytes = (size_t)vertex_ptr[K] * 16;
    slot.o_lam = o_lam;
    slot.lam_bytes = (size_t)K * 24;
    slot.o_w = o_w;
    slot.w_bytes = w_out ? (size_t)w_ptr[K] * 8 : 0;
    *ticket_out = pick;
    return SCS_OK;
}

extern "C" int scs_small_solve_end(scs_ctx *ctx, int32_t ticket, double *maps_out, double *lambda_out,
                                   double *w_out) {
    SCS_REQUIRE(ctx != nullptr, "scs_small_solve_end: null context");
    SCS_REQUIRE(ticket >= 0 && (size_t)ticket < ctx->small_slots.size() && ctx->small_slots[ticket].busy,
                "scs_small_solve_end: no such ticket (%d)", ticket);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    scs_ctx::small_slot &slot = ctx->small_slots[ticket];
    const hipError_t e = hipEventSynchronize(slot.done);
    slot.busy = false;  // (whatever happened: the slot is the caller's no longer)
    if (e != hipSuccess) {
        scs_set_error("scs_small_solve: %s", hipGetErrorString(e));
        return SCS_EHIP;
    }
    if (maps_out) memcpy(maps_out, slot.host + slot.o_maps, slot.maps_bytes);
    if (lambda_out) memcpy(lambda_out, slot.host + slot.o_lam, slot.lam_bytes);
    if (w_out && slot.w_bytes) memcpy(w_out, slot.host + slot.o_w, slot.w_bytes);
    return SCS_OK;
}

extern "C" int scs_small_solve(scs_ctx *ctx, int32_t n_nodes, const int32_t *n_taxa,
                               const int32_t *n_trees, const int32_t *n_groups,
                               const int32_t *tree_off, const int32_t *leaf_taxon,
                               const int32_t *adj_depth, const double *adj_val,
                               const double *tree_w, const int32_t *group_start, double *maps_out,
                               double *lambda_out, double *w_out) {
    SCS_REQUIRE(maps_out && lambda_out, "scs_small_solve: null output");
    int32_t ticket = -1;
    SCS_TRY(scs_small_solve_begin(ctx, n_nodes, n_taxa, n_trees, n_groups, tree_off, leaf_taxon, adj_depth,
                                  adj_val, tree_w, group_start, w_out != nullptr, &ticket));
    return scs_small_solve_end(ctx, ticket, maps_out, lambda_out, w_out);
}

// ---------------------------------------------------------------------------
// debug entry points (parity tests of the building blocks)
// ---------------------------------------------------------------------------
extern "C" int scs_debug_jacobi(scs_ctx *ctx, const double *a, int32_t n, double *w, double *v) {
    SCS_REQUIRE(ctx && a && w && v, "scs_debug_jacobi: null argument");
    SCS_REQUIRE(n >= 1 && n <= MAXS, "scs_debug_jacobi: n must be in [1, %d]", MAXS);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    t_ctx = ctx;
    dbuf da, dw, dv;
    SCS_TRY(da.alloc((size_t)n * n * 8));
    SCS_TRY(dw.alloc((size_t)n * 8));
    SCS_TRY(dv.alloc((size_t)n * n * 8));
    SCS_HIP_CHECK(hipMemcpyAsync(da.p, a, (size_t)n * n * 8, hipMemcpyHostToDevice, ctx->stream));
    k_small_eig<<<1, 256, 0, ctx->stream>>>(da.d(), n, dw.d(), dv.d());
    SCS_HIP_CHECK(hipMemcpyAsync(w, dw.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SCS_HIP_CHECK(hipMemcpyAsync(v, dv.p, (size_t)n * n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SCS_OK;
}

extern "C" int scs_debug_gram(scs_ctx *ctx, const double *a, const double *b, int32_t n,
                              int32_t ka, int32_t kb, int32_t use_mfma, double *out) {
    SCS_REQUIRE(ctx && a && b && out, "scs_debug_gram: null argument");
    SCS_REQUIRE(n >= 1 && ka >= 1 && ka <= 48 && kb >= 1 && kb <= 48, "scs_debug_gram: bad shape");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    t_ctx = ctx;
    solver sv;
    sv.ctx = ctx;
    sv.s = ctx->stream;
    sv.n = n;
    sv.gram_blocks = 256;
    dbuf da, db, dout;
    SCS_TRY(da.alloc((size_t)n * ka * 8));
    SCS_TRY(db.alloc((size_t)n * kb * 8));
    SCS_TRY(dout.alloc((size_t)ka * kb * 8));
    SCS_TRY(sv.part.alloc((size_t)1024 * 48 * 48 * 8));
    SCS_HIP_CHECK(hipMemcpyAsync(da.p, a, (size_t)n * ka * 8, hipMemcpyHostToDevice, ctx->stream));
    SCS_HIP_CHECK(hipMemcpyAsync(db.p, b, (size_t)n * kb * 8, hipMemcpyHostToDevice, ctx->stream));
    SCS_TRY(sv.gram(da.d(), ka, ka, db.d(), kb, kb, dout.d(), use_mfma != 0));
    SCS_HIP_CHECK(hipMemcpyAsync(out, dout.p, (size_t)ka * kb * 8, hipMemcpyDeviceToHost, ctx->stream));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SCS_OK;
}

extern "C" int scs_debug_apply(scs_ctx *ctx, scs_graph *g, const double *x, int32_t b, double *y) {
    SCS_REQUIRE(ctx && g && x && y, "scs_debug_apply: null argument");
    SCS_REQUIRE(b == 4 || b == 8 || b == 12 || b == 16, "scs_debug_apply: b must be 4, 8, 12 or 16");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    t_ctx = ctx;
    SCS_TRY(scs_graph_prepare_degrees(ctx, g));
    solver sv;
    sv.ctx = ctx;
    sv.g = g;
    sv.s = ctx->stream;
    sv.n = g->n;
    sv.b = b;
    sv.rows = g->row_end - g->row_begin;
    sv.world = ctx->comm.world;
    const int n = g->n;
    dbuf dx;
    SCS_TRY(dx.alloc((size_t)n * b * 8));
    SCS_TRY(sv.alloc_symm_buffers());
    // (an SCS_BUILD_UPPER graph: collective, the product comes back for all V rows and this
    // rank's slice of it is returned)
    SCS_TRY(sv.yloc.alloc((size_t)(g->upper ? n : sv.rows) * b * 8));
    SCS_HIP_CHECK(hipMemcpyAsync(dx.p, x, (size_t)n * b * 8, hipMemcpyHostToDevice, ctx->stream));
    k_scale_rows<<<(n * b + 255) / 256, 256, 0, ctx->stream>>>(dx.d(), b, 0, b, n, g->d_dinv,
                                                               sv.z.d(), sv.ldz);
    SCS_TRY(sv.launch_symm(sv.z.d(), sv.yloc.d()));
    SCS_HIP_CHECK(hipMemcpyAsync(y, sv.yloc.d() + (g->upper ? (size_t)g->row_begin * b : 0),
                                 (size_t)sv.rows * b * 8, hipMemcpyDeviceToHost, ctx->stream));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return SCS_OK;
}


// The loop's decision rules on a scripted sequence of residuals (no device): actions_out[i] = 0 go on, 1 stop
// (the NEXT residual of the script is then taken as the confirmation's: the loop ends there or goes on in double
// precision), 2 renew, 3 = ended converged, 4 = ended not converged; entries behind the end are -1.
extern "C" int scs_debug_loop_policy(double tol, int32_t lowp_mode, double lowp_tol, double lowp_tol2,
                                     int32_t image, int32_t n, const double *residuals, int32_t *actions_out) {
    SCS_REQUIRE(residuals && actions_out && n >= 0, "scs_debug_loop_policy: bad arguments");
    scs_loop_policy p;
    p.tol = tol;
    p.lowp_mode = lowp_mode;
    p.lowp_tol = lowp_tol;
    p.lowp_tol2 = lowp_tol2;
    p.lowp_state = image ? 1 : 0;
    bool ended = false;
    for (int32_t i = 0; i < n; ++i) {
        if (ended) {
            actions_out[i] = -1;
            continue;
        }
        const int a = p.step(residuals[i]);
        actions_out[i] = a;
        if (a == scs_loop_policy::STOP) {
            if (!p.begin_confirmation()) {
                actions_out[i] = residuals[i] <= tol ? 3 : 4;
                ended = true;
            } else if (i + 1 < n) {
                bool conv = false;
                ++i;
                if (p.confirm(residuals[i], &conv)) {
                    actions_out[i] = conv ? 3 : 4;
                    ended = true;
                } else {
                    actions_out[i] = 0;
                }
            }
        }
    }
    return SCS_OK;
}
