// k_symm_tri: Y = W Z for a SYMMETRIC W held in full on one device, streaming only the tiles
// on and above the diagonal (about half the bytes of k_symm).
//
// W is cut into tiles of TH = 128 rows x TW = CT * 128 columns.  A workgroup owns one tile
// (I, J) with J >= the column tile that holds the diagonal of row block I and forms
//   direct      pdir[J][r][:]  = sum_{c in tile} W[r][c] z[c][:]        (rows of block I)
//   transposed  ptr[I][c][:]   = sum_{r in tile} W[r][c] z[r][:]        (columns of tile J)
// from the SAME loaded elements; a diagonal tile (the one holding W[r][r] of its rows) is read
// whole and only contributes directly, so every unordered pair is applied exactly twice.
// k_symm_tri_finish adds the partials of a row in a fixed order (deterministic, bitwise
// repeatable): first the direct ones by J, then the transposed ones by I.
//
// A wave owns 32 rows of the tile and walks them RPW (four; two at width 8) at a time with the
// loads of the next D - 1 groups in flight (one 16-byte load per row per lane per 128-column
// sub-chunk).  The direct product needs a sum over the lanes per row: the 4 x B partial
// sums of a group are reduced by recursive halving (lanes exchange half of their values at
// distance 32, 16, ...: 4 B + ... shuffles instead of 6 x 4 B); the transposed product
// accumulates down the columns in registers and is combined across the four waves once per
// tile.  Layout contract as k_symm: ld a multiple of 512 doubles, padding columns of W and of
// zt zero (rows beyond n are clamped to a real row and meet z = 0).  zt has its own leading
// dimension ldz: in a multi-rank SCS_BUILD_UPPER graph a rank's W holds only the columns from its
// first row on (w then points col0 columns and row_begin rows in front of the stored block, so
// that global indices address it), and `tiles` lists the tiles of the rank's own row blocks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int TRI_TH = 128;  // rows per tile

// sum NV (8, 16 or 32) per-lane values over the 64 lanes by recursive halving; on return the
// lanes that are multiples of 64 / NV hold in v[0] the total of value index lane / (64 / NV)
template <int NV>
__device__ __forceinline__ void tri_halving_reduce(double (&v)[NV], int lane) {
    int c = NV;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        if (c > 1) {
            const bool up = (lane & off) != 0;
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) {
                if (j < c / 2) {
                    const double lo = v[j], hi = v[j + c / 2];
                    const double send = up ? lo : hi;
                    const double keep = up ? hi : lo;
                    v[j] = keep + __shfl_xor(send, off, 64);
                }
            }
            c /= 2;
        } else {
            v[0] += __shfl_xor(v[0], off, 64);
        }
    }
}

// A lane's 16-byte piece of a row of W: two doubles, or four floats of the single-precision image W32
// (round 5: the operator applied to the search directions inside the LOBPCG loop; scs_eig.hip)
template <typename WT>
struct tri_piece;
template <>
struct tri_piece<double> {
    static constexpr int N = 2;
    double2 v;
    __device__ __forceinline__ double at(int j) const { return j == 0 ? v.x : v.y; }
};
template <>
struct tri_piece<float> {
    static constexpr int N = 4;
    float4 v;
    __device__ __forceinline__ double at(int j) const {
        return (double)(j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w);
    }
};

// LDS of one workgroup of the symmetric SYMM: B x (2 TW + 128) doubles (TW = CT sub-chunks of 64 pieces)
template <int B, int CT, typename WT = double>
struct symm_tri_lds {
    static constexpr int TW = CT * 64 * tri_piece<WT>::N;
    alignas(16) double zc[B][TW];
    alignas(16) double zr[B][TRI_TH];
    alignas(16) double red[B][TW];
};

template <int B, int CT, int RPW, int D, typename WT = double>
__device__ __forceinline__ void symm_tri_body(const WT *__restrict__ w, int64_t ld, int n,
                                              const double *__restrict__ zt, int64_t ldz, const int2 tile,
                                              double *__restrict__ pdir, double *__restrict__ ptr_,
                                              int64_t panel_stride, symm_tri_lds<B, CT, WT> &lds) {
    // panel_stride > 0 (tools/symm_tri_bench.hip only: a measurement of what a tile-major storage of W
    // would buy): column panel J of TW columns is stored on its own, row-major with leading dimension
    // TW, at w + J * panel_stride -- a tile is then one contiguous piece of memory
    static_assert(B == 4 || B == 8, "block widths 4 and 8");
    typedef tri_piece<WT> piece;
    constexpr int PN = piece::N;       // columns per lane and sub-chunk
    constexpr int SUB = 64 * PN;       // columns per sub-chunk (1 KB of a row)
    constexpr int TW = CT * SUB;
    constexpr int NG = 32 / RPW;  // groups of RPW rows per wave: 32 rows
    constexpr int NV = RPW * B;
    static_assert(NV == 8 || NV == 16 || NV == 32, "partial sums per group");
    double (&zc)[B][TW] = lds.zc;
    double (&zr)[B][TRI_TH] = lds.zr;
    double (&red)[B][TW] = lds.red;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = tile.x * TRI_TH, cb = tile.y * TW;
    const bool diag = tile.y == rb / TW;  // uniform over the workgroup

    // ---- operands into LDS (16-byte pieces)
    for (int e = tid * 2; e < B * TW; e += 512) {
        const int k = e / TW, c = e - k * TW;
        *(double2 *)&zc[k][c] = *(const double2 *)(zt + (int64_t)k * ldz + cb + c);
    }
    for (int e = tid * 2; e < B * TRI_TH; e += 512) {
        const int k = e / TRI_TH, r = e - k * TRI_TH;
        *(double2 *)&zr[k][r] = *(const double2 *)(zt + (int64_t)k * ldz + rb + r);
    }

    const int r_wave = rb + wave * (RPW * NG);
    auto row_ptr = [&](int g, int i) {
        int r = r_wave + g * RPW + i;
        r = r < n ? r : n - 1;  // clamped rows meet z = 0 (transposed) and are not stored (direct)
        if (panel_stride) return (const char *)(w + (int64_t)tile.y * panel_stride + (int64_t)r * TW) + lane * 16;
        return (const char *)(w + (int64_t)r * ld + cb) + lane * 16;
    };
    // D pipeline stages: the loads of D - 1 groups are in flight while one is used
    piece a[D][RPW][CT];
#pragma unroll
    for (int g = 0; g < D - 1; ++g)
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int s = 0; s < CT; ++s) a[g][i][s].v = *(const decltype(piece::v) *)(row_ptr(g, i) + s * 1024);

    double acct[CT][PN][B];
#pragma unroll
    for (int s = 0; s < CT; ++s)
#pragma unroll
        for (int j = 0; j < PN; ++j)
#pragma unroll
            for (int k = 0; k < B; ++k) acct[s][j][k] = 0.0;
    __syncthreads();

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int cur = g % D;
        if (g + D - 1 < NG) {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int s = 0; s < CT; ++s)
                    a[(g + D - 1) % D][i][s].v = *(const decltype(piece::v) *)(row_ptr(g + D - 1, i) + s * 1024);
        }
        double acc[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] = 0.0;
#pragma unroll
        for (int s = 0; s < CT; ++s) {
#pragma unroll
            for (int k = 0; k < B; ++k) {
                // (loop-invariant over the unrolled groups: the compiler keeps these in registers)
                double zz[PN];
#pragma unroll
                for (int j = 0; j < PN; j += 2) {
                    const double2 t = *(const double2 *)&zc[k][s * SUB + PN * lane + j];
                    zz[j] = t.x;
                    zz[j + 1] = t.y;
                }
#pragma unroll
                for (int i = 0; i < RPW; ++i)
#pragma unroll
                    for (int j = 0; j < PN; ++j) acc[i * B + k] = fma(a[cur][i][s].at(j), zz[j], acc[i * B + k]);
            }
        }
        if (!diag) {
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int rl = wave * (RPW * NG) + g * RPW + i;
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const double zz = zr[k][rl];  // one address for the whole wave: broadcast
#pragma unroll
                    for (int s = 0; s < CT; ++s)
#pragma unroll
                        for (int j = 0; j < PN; ++j) acct[s][j][k] = fma(a[cur][i][s].at(j), zz, acct[s][j][k]);
                }
            }
        }
        tri_halving_reduce<NV>(acc, lane);
        {
            constexpr int LOW = 64 / NV;  // lanes per held value
            const int idx = lane / LOW;
            const int r = r_wave + g * RPW + idx / B;
            if ((lane & (LOW - 1)) == 0 && r < n)
                pdir[((int64_t)tile.y * n + r) * B + (idx % B)] = acc[0];
        }
    }
    if (!diag) {
        // the four waves' column sums, added up in wave order (one LDS image, four turns)
#pragma unroll
        for (int turn = 0; turn < 4; ++turn) {
            if (wave == turn) {
#pragma unroll
                for (int s = 0; s < CT; ++s)
#pragma unroll
                    for (int k = 0; k < B; ++k)
#pragma unroll
                        for (int j = 0; j < PN; j += 2) {
                            double2 *slot = (double2 *)&red[k][s * SUB + PN * lane + j];
                            double2 v = make_double2(acct[s][j][k], acct[s][j + 1][k]);
                            if (turn > 0) {
                                const double2 old = *slot;
                                v.x = old.x + v.x;
                                v.y = old.y + v.y;
                            }
                            *slot = v;
                        }
            }
            __syncthreads();
        }
        for (int e = tid; e < B * TW; e += 256) {
            const int c = e / B, k = e - c * B;
            if (cb + c < n) ptr_[((int64_t)tile.x * n + cb + c) * B + k] = red[k][c];
        }
    }
}

template <int B, int CT, int RPW, int D, typename WT = double>
__global__ __launch_bounds__(256, 2) void k_symm_tri(const WT *__restrict__ w, int64_t ld, int n,
                                                     const double *__restrict__ zt, int64_t ldz,
                                                     const int2 *__restrict__ tiles,
                                                     double *__restrict__ pdir,
                                                     double *__restrict__ ptr_,
                                                     int64_t panel_stride = 0) {
    __shared__ symm_tri_lds<B, CT, WT> lds;
    symm_tri_body<B, CT, RPW, D, WT>(w, ld, n, zt, ldz, tiles[blockIdx.x], pdir, ptr_, panel_stride, lds);
}

// y[r][:] = scale(r) * ( sum_J pdir[J][r][:] + sum_I ptr[I][r][:] ), J from the diagonal tile
// of r's row block upwards, I over the row blocks left of r's column tile; fixed order.
// dinv == nullptr: no scaling (the fused LOBPCG loop applies it when it folds the result in).
// rb_lo, rb_hi: the row blocks (of TRI_TH rows) whose tiles this rank streamed -- all of them on
// a single device; a rank of an SCS_BUILD_UPPER job adds the direct partials only for its own
// rows and the transposed ones only from its own row blocks: its y is then a PARTIAL product
// (the ranks' vectors are added in rank order after the all-gather).
__global__ __launch_bounds__(256) void k_symm_tri_finish(const double *__restrict__ pdir,
                                                           const double *__restrict__ ptr_, int n, int b,
                                                           int tw, int n_ct,
                                                           const double *__restrict__ dinv,
                                                           double *__restrict__ y, int rb_lo = 0,
                                                           int rb_hi = 0x7FFFFFFF) {
    // Four lanes share an output: lane q of the quad sums the partials q, q + 4, ... (eight
    // independent loads in flight at a time), the quad is combined in a fixed order.
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int idx = gid >> 2, q = gid & 3;
    const bool live = idx < n * b;
    const int r = live ? idx / b : 0;
    const int j0 = (r / TRI_TH) * TRI_TH / tw;  // the diagonal tile of r's row block
    const int i1 = min((r / tw) * tw / TRI_TH, rb_hi);  // row blocks strictly left of r's column tile
    const int i0 = min(rb_lo, i1);
    const bool mine = r / TRI_TH >= rb_lo && r / TRI_TH < rb_hi;
    const int n_dir = mine ? n_ct - j0 : 0, n_all = n_dir + (i1 - i0);
    const int64_t stride = (int64_t)n * b;
    double s = 0.0;
    if (live) {
        for (int t0 = q; t0 < n_all; t0 += 32) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 4 * u;
                v[u] = 0.0;
                if (t < n_dir) v[u] = pdir[(int64_t)(j0 + t) * stride + idx];
                else if (t < n_all) v[u] = ptr_[(int64_t)(i0 + t - n_dir) * stride + idx];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    }
    const double s1 = __shfl_xor(s, 1, 64);
    const double pair = (q & 1) ? s1 + s : s + s1;      // (q0 + q1), (q2 + q3): same value on both lanes
    const double other = __shfl_xor(pair, 2, 64);
    const double total = (q & 2) ? other + pair : pair + other;
    if (live && q == 0) y[idx] = dinv ? dinv[r] * total : total;
}
