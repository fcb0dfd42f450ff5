// Fused panel kernels of the LOBPCG iteration (block widths 4 and 8).
//
// Everything in an iteration except k_symm works on N x 3b panels Q = [X | P | R] and
// AQ = S Q, a megabyte or two that lives in L2: those steps are bound by kernel-launch
// and dependency latency, not bandwidth, so an iteration is cut into as few launches as
// the data dependencies allow:
//
//   k_panel_rr        X' = Q c, P' = Q d, AX' = AQ c, AP' = AQ d, R = AX' - X' theta,
//                     and the Gram products [u X' P']^T R, R^T R          (one pass over Q, AQ)
//   k_small_orth      projected SVQB: from those Gram products the transform that makes R
//                     orthonormal and orthogonal to [u X P]  (also hands the residual norms to
//                     the host)
//   k_panel_tf        R <- R T + [u X P] K, again with the Gram products     (second pass)
//   k_small_orth
//   k_panel_tf        final transform, also emits Z = D^-1/2 R k-major for k_symm
//   k_symm
//   k_gram_qaq        combines k_symm's column segments into AR and forms Q^T AQ
//   k_small_rr        Rayleigh-Ritz (scs_eig.hip)
//
// Since round 4 the three small solves run in FRONT of the tall kernel that consumes them, in
// every one of its workgroups (k_panel_rr_solve, k_panel_tf_solve in scs_eig.hip: same partials,
// same code, same bits everywhere; workgroup 0 publishes what later kernels and the host read):
// six launches an iteration instead of nine, 17.8 -> 17.2 ms per step at 10 000 vertices.
// SCS_SPLIT_SMALL=1 keeps them as one-workgroup kernels of their own.
// (Running a small solve in the TAIL of the tall kernel before it -- the workgroup whose
// agent-scope ticket add comes last sums the partials and solves -- was tried in round 2:
// bit-identical results, but no faster: 10 000 vertices 14.0 ms per solve against 13.6 ms,
// 1 000 vertices 4.97 against 5.04 ms.  Every workgroup's release has to write its XCD's L2
// back, which costs what the launch saved; the redundant solve in front shares nothing.)
//
// The tall kernels are MFMA pipelines (v_mfma_f64_16x16x4_f64): a wave owns groups of 16
// rows; the row transform is a 16 x K x 16 product whose accumulator layout (register r
// of lane l = row (l>>4) + 4r, column l&15) is exactly the operand layout of the Gram
// product over four rows, so transformed rows feed the Gram MFMAs straight from
// registers.  All sums are formed in a fixed order (per-wave accumulators, workgroup
// partials, then one thread per output in the small kernel): results are deterministic.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef double panel_v4d __attribute__((ext_vector_type(4)));

constexpr int PANEL_BLOCKS_MAX = 128;  // workgroups of a tall panel kernel (partials per output; 64 -> 128: -3 % of the solve)

// outputs per workgroup partial: [x p]^T r (2B x B), then [r | u]^T r ((B + 1) x B)
template <int B>
struct panel_out {
    static constexpr int XP = 2 * B * B;
    static constexpr int RU = (B + 1) * B;
    static constexpr int TOTAL = XP + RU;
};

// coefficient matrix of k_panel_tf: rows [x (B) | p (B) | r (B) | u | 3 x pad], B columns
template <int B>
constexpr int PANEL_COEF_ROWS = 3 * B + 4;

typedef double panel_red_t[4][2][4][64];

template <int B>
__device__ __forceinline__ void panel_store_partials(panel_v4d g1, panel_v4d g2,
                                                     double *__restrict__ partial, int bid = -1,
                                                     panel_red_t *red_ext = nullptr) {
    // cross-wave sum in fixed order through LDS, then one partial per workgroup
    // (red_ext: the caller's LDS -- a kernel that shares its LDS between two roles, scs_eig.hip)
    __shared__ panel_red_t red_own;
    panel_red_t &red = red_ext ? *red_ext : red_own;
    if (bid < 0) bid = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][0][r][lane] = g1[r];
        red[wave][1][r][lane] = g2[r];
    }
    __syncthreads();
    if (wave == 0) {
        double *out = partial + (int64_t)bid * panel_out<B>::TOTAL;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = kk + 4 * r;  // output row
            const double s1 = ((red[0][0][r][lane] + red[1][0][r][lane]) + red[2][0][r][lane]) +
                              red[3][0][r][lane];
            const double s2 = ((red[0][1][r][lane] + red[1][1][r][lane]) + red[2][1][r][lane]) +
                              red[3][1][r][lane];
            if (i < 2 * B && cc < B) out[i * B + cc] = s1;
            if (i <= B && cc < B) out[panel_out<B>::XP + i * B + cc] = s2;
        }
    }
}

// Rayleigh-Ritz update + residual + Gram products.  q, aq: n x 3B (ld 3B), updated in
// place (a group of 16 rows is read completely by its wave before it is written).
// c, d: 3B x B row-major; theta: B; u: n or null.
// (c, d, theta may live in LDS: the solve-in-front kernels of scs_eig.hip pass their own)
template <int B>
__device__ __forceinline__ void panel_rr_body(double *q, double *aq, const double *__restrict__ u,
                                              const double *c, const double *d, const double *theta,
                                              int n, double *__restrict__ partial,
                                              const double *__restrict__ dinv = nullptr,
                                              double *__restrict__ zt = nullptr, int64_t ldz = 0) {
    static_assert(B == 4 || B == 8, "fused panel kernels take block widths 4 and 8");
    constexpr int LD = 3 * B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    const int n_groups = (n + 15) / 16;
    panel_v4d g1 = {0.0, 0.0, 0.0, 0.0}, g2 = {0.0, 0.0, 0.0, 0.0};
    const double th = cc < B ? theta[cc] : 0.0;
    // B operand of the transform: coefficient rows k0 + kk, column cc of [c | d]
    double coef[LD / 4];
#pragma unroll
    for (int k = 0; k < LD / 4; ++k) {
        const int row = 4 * k + kk;
        coef[k] = cc < B ? c[row * B + cc] : (cc < 2 * B ? d[row * B + cc - B] : 0.0);
    }
    for (int grp = blockIdx.x * 4 + wave; grp < n_groups; grp += gridDim.x * 4) {
        const int base = grp * 16;
        const int ra = base + cc;  // row of this lane's A operand
        panel_v4d dq = {0.0, 0.0, 0.0, 0.0}, da = {0.0, 0.0, 0.0, 0.0};
        double aq_op[LD / 4], q_op[LD / 4];
#pragma unroll
        for (int k = 0; k < LD / 4; ++k) {
            q_op[k] = ra < n ? q[(int64_t)ra * LD + 4 * k + kk] : 0.0;
            aq_op[k] = ra < n ? aq[(int64_t)ra * LD + 4 * k + kk] : 0.0;
        }
        double u_in[4];  // (read before the stores below: see k_panel_tf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = base + kk + 4 * r;
            u_in[r] = (cc == B && u && row < n) ? u[row] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < LD / 4; ++k) {
            dq = __builtin_amdgcn_mfma_f64_16x16x4f64(q_op[k], coef[k], dq, 0, 0, 0);
            da = __builtin_amdgcn_mfma_f64_16x16x4f64(aq_op[k], coef[k], da, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = base + kk + 4 * r;
            const bool live = row < n;
            const double res = cc < B ? da[r] - dq[r] * th : 0.0;
            double ru = res;
            if (cc == B) ru = u_in[r];
            if (live && cc < 2 * B) {
                q[(int64_t)row * LD + cc] = dq[r];
                aq[(int64_t)row * LD + cc] = da[r];
            }
            if (live && cc < B) {
                q[(int64_t)row * LD + 2 * B + cc] = res;
                // (round 5, overlapped loop: the operator is applied to the RAW residual block)
                if (zt) zt[(int64_t)cc * ldz + row] = dinv[row] * res;
            }
            g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(dq[r], res, g1, 0, 0, 0);
            g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ru, res, g2, 0, 0, 0);
        }
    }
    panel_store_partials<B>(g1, g2, partial);
}

template <int B>
__global__ __launch_bounds__(256) void k_panel_rr(double *q, double *aq,
                                                   const double *__restrict__ u,
                                                   const double *__restrict__ c,
                                                   const double *__restrict__ d,
                                                   const double *__restrict__ theta, int n,
                                                   double *__restrict__ partial) {
    panel_rr_body<B>(q, aq, u, c, d, theta, n, partial);
}

// R <- R T + [u X P] K (coefficients: PANEL_COEF_ROWS<B> x B, rows [x | p | r | u | pad]).
// GRAM: also the Gram products of the new R.  WRITE_Z: also zt[k][row] = dinv[row] R[row][k].
template <int B, bool GRAM, bool WRITE_Z>
__device__ __forceinline__ void panel_tf_body(double *q, const double *__restrict__ u,
                                              const double *coefm, int n,
                                              double *__restrict__ partial,
                                              const double *__restrict__ dinv,
                                              double *__restrict__ zt, int64_t ldz, int bid = -1,
                                              int nblk = -1, panel_red_t *red_ext = nullptr) {
    static_assert(B == 4 || B == 8, "fused panel kernels take block widths 4 and 8");
    constexpr int LD = 3 * B;
    constexpr int KC = PANEL_COEF_ROWS<B> / 4;
    if (bid < 0) {
        bid = blockIdx.x;
        nblk = gridDim.x;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    const int n_groups = (n + 15) / 16;
    panel_v4d g1 = {0.0, 0.0, 0.0, 0.0}, g2 = {0.0, 0.0, 0.0, 0.0};
    double coef[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) coef[k] = cc < B ? coefm[(4 * k + kk) * B + cc] : 0.0;
    for (int grp = bid * 4 + wave; grp < n_groups; grp += nblk * 4) {
        const int base = grp * 16;
        const int ra = base + cc;
        double a_op[KC];
#pragma unroll
        for (int k = 0; k < KC - 1; ++k) a_op[k] = ra < n ? q[(int64_t)ra * LD + 4 * k + kk] : 0.0;
        a_op[KC - 1] = (kk == 0 && u && ra < n) ? u[ra] : 0.0;
        // the [x p] entries of the Gram products, read with the operands: after the stores of
        // the new R below a load from q would have to wait for them (they cannot alias, the
        // compiler cannot know) -- a memory round trip per four rows in a kernel that is all latency
        double xp_in[4], u_in[4];
        if (GRAM) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = base + kk + 4 * r;
                xp_in[r] = (row < n && cc < 2 * B) ? q[(int64_t)row * LD + cc] : 0.0;
                u_in[r] = (cc == B && u && row < n) ? u[row] : 0.0;
            }
        }
        panel_v4d dr = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < KC; ++k)
            dr = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[k], coef[k], dr, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = base + kk + 4 * r;
            const bool live = row < n;
            const double res = dr[r];  // columns >= B are zero (zero coefficients)
            if (live && cc < B) {
                q[(int64_t)row * LD + 2 * B + cc] = res;
                if (WRITE_Z) zt[(int64_t)cc * ldz + row] = dinv[row] * res;
            }
            if (GRAM) {
                const double xp = xp_in[r];
                double ru = res;
                if (cc == B) ru = u_in[r];
                g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xp, res, g1, 0, 0, 0);
                g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ru, res, g2, 0, 0, 0);
            }
        }
    }
    if (GRAM) panel_store_partials<B>(g1, g2, partial, bid, red_ext);
}

template <int B, bool GRAM, bool WRITE_Z>
__global__ __launch_bounds__(256) void k_panel_tf(double *q, const double *__restrict__ u,
                                                   const double *__restrict__ coefm, int n,
                                                   double *__restrict__ partial,
                                                   const double *__restrict__ dinv,
                                                   double *__restrict__ zt, int64_t ldz) {
    panel_tf_body<B, GRAM, WRITE_Z>(q, u, coefm, n, partial, dinv, zt, ldz);
}

// T = Q^T AQ (3B x 3B) as per-workgroup partials.  FINISH selects where AR, the R slot of AQ,
// comes from -- it is stored into AQ on the way:
//   1  single-rank runs: combine k_symm's column segments, AR = dinv (.) sum_seg ypart[seg];
//   2  multi-rank runs: the all-gathered row slices (ypart = the receive buffer: `nseg` ranks x
//      `chunk` doubles, rank r's rows [splits[r], splits[r+1]), already scaled);
//   3  SCS_BUILD_UPPER jobs: ypart = the gathered PARTIAL products of the `nseg` ranks (each
//      n x B, unscaled): AR = dinv (.) their sum in rank order;
//   0  AR is in place.
template <int B, int FINISH>
__global__ __launch_bounds__(256) void k_gram_qaq(const double *__restrict__ q, double *aq, int n,
                                                   const double *__restrict__ ypart, int nseg,
                                                   const double *__restrict__ dinv,
                                                   double *__restrict__ partial,
                                                   int64_t chunk = 0,
                                                   const int32_t *__restrict__ splits = nullptr) {
    static_assert(B == 4 || B == 8, "fused panel kernels take block widths 4 and 8");
    constexpr int LD = 3 * B;
    constexpr int NT = (LD + 15) / 16;  // 16-column tiles of the panels
    __shared__ double red[4][NT * NT][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    const double *aq_in = aq;
    panel_v4d acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (panel_v4d){0.0, 0.0, 0.0, 0.0};
    const int n_groups = (n + 3) / 4;
    // U groups of four rows per trip: all their loads are issued before the first store and
    // the first MFMA, so the trips of a wave are bound by one memory latency, not U
    constexpr int U = 4;
    const int stride = gridDim.x * 4;
    for (int grp0 = blockIdx.x * 4 + wave; grp0 < n_groups; grp0 += stride * U) {
        double fa[U][NT], fb[U][NT];
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            const int row = (grp0 + uu * stride) * 4 + kk;
            const bool live = grp0 + uu * stride < n_groups && row < n;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + cc;
                fa[uu][t] = (live && col < LD) ? q[(int64_t)row * LD + col] : 0.0;
                double v = 0.0;
                if (live && col < LD) {
                    if (FINISH == 2 && col >= 2 * B) {
                        int r = 0;
                        while (r + 1 < nseg && row >= splits[r + 1]) ++r;
                        v = ypart[(int64_t)r * chunk + (int64_t)(row - splits[r]) * B + (col - 2 * B)];
                    } else if (FINISH == 3 && col >= 2 * B) {
                        const int64_t idx = (int64_t)row * B + (col - 2 * B);
                        double sum = 0.0;
                        for (int g = 0; g < nseg; ++g) sum += ypart[(int64_t)g * n * B + idx];
                        v = dinv[row] * sum;
                    } else if (FINISH == 1 && col >= 2 * B) {
                        const int64_t idx = (int64_t)row * B + (col - 2 * B);
                        // k_symm uses at most four column segments; the loads are independent
                        double part[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            part[g] = g < nseg ? ypart[(int64_t)g * n * B + idx] : 0.0;
                        const double sum = ((part[0] + part[1]) + part[2]) + part[3];
                        v = dinv[row] * sum;
                    } else {
                        v = aq_in[(int64_t)row * LD + col];
                    }
                }
                fb[uu][t] = v;
            }
        }
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            const int row = (grp0 + uu * stride) * 4 + kk;
            const bool live = grp0 + uu * stride < n_groups && row < n;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + cc;
                if (FINISH && live && col < LD && col >= 2 * B) aq[(int64_t)row * LD + col] = fb[uu][t];
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[uu][i], fb[uu][j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i * NT + j][r][lane] = acc[i][j][r];
    __syncthreads();
    if (wave == 0) {
        double *out = partial + (int64_t)blockIdx.x * LD * LD;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = i * NT + j;
                    const double s = ((red[0][t][r][lane] + red[1][t][r][lane]) + red[2][t][r][lane]) +
                                     red[3][t][r][lane];
                    const int orow = i * 16 + kk + 4 * r, ocol = j * 16 + cc;
                    if (orow < LD && ocol < LD) out[orow * LD + ocol] = s;
                }
    }
}

// ---------------------------------------------------------------------------
// Round 5: the SECOND orthonormalisation pass of R folded into the Gram kernel.
// The iteration used to be  rr -> tf (pass 1) -> tf (pass 2, writes Z) -> SYMM -> finish -> gram.
// Pass 2 only removes the rounding left by pass 1 (its transform is the identity to ~1e-16), and
// it is linear: R2 = [X P R1 u] K  implies  S R2 = [SX SP SR1 u] K  (S u = u).  So pass 1 writes Z,
// the operator is applied to R1, and THIS kernel -- which combines the SYMM partials into S R1 and
// forms Q^T (S Q) anyway -- applies K to both R1 and S R1 on the way, for its rows, before the Gram
// products: one tall launch and one small solve fewer on the critical path of every iteration.
// `coefm`: PANEL_COEF_ROWS<B> x B, rows [x | p | r | u | pad] (small_orth_body's output).
// Tiles of 16 rows per wave as panel_rr_body: the transform's accumulator layout is the Gram
// product's operand layout, and the new R columns land where the Gram product wants them
// (columns 2B .. 3B-1 of the panel) because the coefficient columns are shifted there.
// ---------------------------------------------------------------------------
template <int B, int FINISH>
__device__ __forceinline__ void panel_gram_tf_body(double *q, double *aq, const double *__restrict__ u,
                                                   const double *coefm, const double *coefam, int n,
                                                   const double *__restrict__ ypart, int nseg,
                                                   const double *__restrict__ dinv,
                                                   double *__restrict__ partial, int64_t chunk,
                                                   const int32_t *__restrict__ splits) {
    static_assert(B == 4 || B == 8, "fused panel kernels take block widths 4 and 8");
    constexpr int LD = 3 * B;
    constexpr int KC = PANEL_COEF_ROWS<B> / 4;
    constexpr int NT = (LD + 15) / 16;     // 16-column tiles of the panels
    constexpr int RT = (2 * B) / 16;       // tile that holds the R columns (B = 4: tile 0; B = 8: tile 1)
    constexpr int RC0 = 2 * B - 16 * RT;   // first R column inside that tile
    __shared__ double red[4][NT * NT][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cc = lane & 15;
    const int n_groups = (n + 15) / 16;
    panel_v4d acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (panel_v4d){0.0, 0.0, 0.0, 0.0};
    // B operand of the transform: coefficient rows 4k + kk, output column cc of tile RT = R column cc - RC0
    // (coefam: the coefficients for the S Q side when they differ -- the overlapped loop applied the
    // operator to the raw residual block, so S R2 comes from [SX SP S R~ u] through both passes' transforms)
    double coef[KC], coefa[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const bool rc = cc >= RC0 && cc < RC0 + B;
        coef[k] = rc ? coefm[(4 * k + kk) * B + (cc - RC0)] : 0.0;
        coefa[k] = rc ? coefam[(4 * k + kk) * B + (cc - RC0)] : 0.0;
    }
    for (int grp = blockIdx.x * 4 + wave; grp < n_groups; grp += gridDim.x * 4) {
        const int base = grp * 16;
        const int ra = base + cc;  // row of this lane's A operand
        // ---- operands of the two transforms: [x p r u] rows of Q, [Sx Sp Sr u] rows of SQ
        double q_op[KC], a_op[KC];
#pragma unroll
        for (int k = 0; k < KC - 1; ++k) {
            const int col = 4 * k + kk;
            q_op[k] = ra < n ? q[(int64_t)ra * LD + col] : 0.0;
            double v = 0.0;
            if (ra < n) {
                if (col < 2 * B || FINISH == 0) {
                    v = aq[(int64_t)ra * LD + col];
                } else if (FINISH == 1) {
                    const int64_t idx = (int64_t)ra * B + (col - 2 * B);
                    double part[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) part[g] = g < nseg ? ypart[(int64_t)g * n * B + idx] : 0.0;
                    v = dinv[ra] * (((part[0] + part[1]) + part[2]) + part[3]);
                } else if (FINISH == 2) {
                    int r = 0;
                    while (r + 1 < nseg && ra >= splits[r + 1]) ++r;
                    v = ypart[(int64_t)r * chunk + (int64_t)(ra - splits[r]) * B + (col - 2 * B)];
                } else {
                    const int64_t idx = (int64_t)ra * B + (col - 2 * B);
                    double sum = 0.0;
                    for (int g = 0; g < nseg; ++g) sum += ypart[(int64_t)g * n * B + idx];
                    v = dinv[ra] * sum;
                }
            }
            a_op[k] = v;
        }
        {
            const double uu = (kk == 0 && u && ra < n) ? u[ra] : 0.0;
            q_op[KC - 1] = uu;
            a_op[KC - 1] = uu;  // S u = u
        }
        // ---- the x, p columns in the Gram product's layout (read before the stores below)
        double xq[NT][4], xa[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = base + kk + 4 * r, col = t * 16 + cc;
                const bool live = row < n && col < 2 * B;
                xq[t][r] = live ? q[(int64_t)row * LD + col] : 0.0;
                xa[t][r] = live ? aq[(int64_t)row * LD + col] : 0.0;
            }
        panel_v4d dr = {0.0, 0.0, 0.0, 0.0}, da = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            dr = __builtin_amdgcn_mfma_f64_16x16x4f64(q_op[k], coef[k], dr, 0, 0, 0);
            da = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[k], coefa[k], da, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = base + kk + 4 * r;
            const bool rcol = cc >= RC0 && cc < RC0 + B;
            if (row < n && rcol) {
                q[(int64_t)row * LD + 2 * B + (cc - RC0)] = dr[r];
                aq[(int64_t)row * LD + 2 * B + (cc - RC0)] = da[r];
            }
            // tile RT of the panels: x / p columns from memory, R columns from the transform
            double fq[NT], fa[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fq[t] = xq[t][r];
                fa[t] = xa[t][r];
            }
            if (rcol) {
                fq[RT] = row < n ? dr[r] : 0.0;
                fa[RT] = row < n ? da[r] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fq[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i * NT + j][r][lane] = acc[i][j][r];
    __syncthreads();
    if (wave == 0) {
        double *out = partial + (int64_t)blockIdx.x * LD * LD;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = i * NT + j;
                    const double sres = ((red[0][t][r][lane] + red[1][t][r][lane]) + red[2][t][r][lane]) +
                                        red[3][t][r][lane];
                    const int orow = i * 16 + kk + 4 * r, ocol = j * 16 + cc;
                    if (orow < LD && ocol < LD) out[orow * LD + ocol] = sres;
                }
    }
}
