// k_symm: the one kernel that streams the N x N matrix (HBM-bound).
//
//   ypart[seg][r][:] = sum_{j in column segment seg} W[r][j] * z[j][:]
//
// Layout contract (no bounds checks and no data-dependent branches in the loop):
//   * W rows have a leading dimension ld that is a multiple of SYMM_LD_ALIGN
//     doubles and the padding columns [n, ld) hold zeros;
//   * zt = (D^-1/2 X)^T is (B x ld), k-major, with columns [n, ld) zero;
//   * B is a multiple of 4.
// zt is staged through a double-buffered LDS tile of S*128 columns in the same
// [k][col] layout (straight 16-byte copies in; a lane's two columns are one
// 16-byte, conflict-free read out) shared by the four waves of the workgroup.
// Every wave streams RPW rows of W, one 16-byte load per row per lane per
// 128-column sub-chunk, software-pipelined S sub-chunks deep in registers
// (RPW*S KiB of W in flight per wave), one workgroup barrier per S sub-chunks.
// The 64 lanes are reduced with wave shuffles at the end.  Column segments
// (gridDim.y) give small problems enough workgroups; partials are combined in
// fixed order by k_symm_finish (deterministic), which also applies the row
// scaling.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int SYMM_SUB = 128;       // columns per sub-chunk (two per lane)
constexpr int SYMM_LD_ALIGN = 512;  // leading dimension granularity (doubles)

// WT = float (round 6): the same launch on the SINGLE-PRECISION IMAGE of a rank's rows (row-partitioned jobs:
// the mixed-precision LOBPCG loop of scs_eig.hip, until round 5 one-device only) -- a lane's two columns are one
// 8-byte load, every product and sum stays in double precision, half the bytes streamed.
template <typename WT>
struct symm_pair;
template <>
struct symm_pair<double> {
    typedef double2 type;
};
template <>
struct symm_pair<float> {
    typedef float2 type;
};

template <int B, int RPW, int S, int MINW = 2, typename WT = double>
__global__ __launch_bounds__(256, MINW) void k_symm(const WT *__restrict__ w, int64_t ld,
                                                     int rows, const double *__restrict__ zt,
                                                     double *__restrict__ ypart,
                                                     int macros_per_seg) {
    typedef typename symm_pair<WT>::type wpair;
    constexpr int WB = (int)sizeof(WT);  // bytes per element of W
    static_assert(B % 4 == 0, "block width must be a multiple of 4");
    static_assert(SYMM_LD_ALIGN % (S * SYMM_SUB) == 0, "S*128 must divide the ld granularity");
    constexpr int MC = S * SYMM_SUB;  // columns per macro-chunk
    constexpr int ZPT = B / 4;        // 16-byte z pieces per thread per sub-chunk
    __shared__ double zs[2][B][MC];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = (blockIdx.x * 4 + wave) * RPW;
    const int n_macros = (int)(ld / MC);
    const int m_begin = blockIdx.y * macros_per_seg;
    const int m_end = min(n_macros, m_begin + macros_per_seg);
    if (m_begin >= m_end) return;  // uniform over the workgroup

    const char *rowp[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        int r = r0 + i;
        r = r < rows ? r : rows - 1;  // clamped rows are computed but never stored
        rowp[i] = (const char *)(w + (int64_t)r * ld) + lane * 2 * WB;
    }
    // z piece q of this thread: k = tid/64 + 4q, column pair tid%64
    const char *zp = (const char *)zt + ((int64_t)(tid >> 6) * ld + 2 * (tid & 63)) * 8;
    const int64_t zstep = 4 * ld * 8;  // bytes between successive pieces (4 k-rows)

    double acc[RPW][B];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int k = 0; k < B; ++k) acc[i][k] = 0.0;

    wpair a[S][RPW];
    double2 zr[ZPT];

    // ---- prologue: first macro-chunk into registers / LDS buffer 0
    {
        const int64_t cb = (int64_t)m_begin * MC * 8;
        const int64_t cbw = (int64_t)m_begin * MC * WB;
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int q = 0; q < ZPT; ++q)
                zr[q] = *(const double2 *)(zp + cb + s * (SYMM_SUB * 8) + q * zstep);
#pragma unroll
            for (int i = 0; i < RPW; ++i)
                a[s][i] = *(const wpair *)(rowp[i] + cbw + s * (SYMM_SUB * WB));
#pragma unroll
            for (int q = 0; q < ZPT; ++q)
                *(double2 *)&zs[0][(tid >> 6) + 4 * q][s * SYMM_SUB + 2 * (tid & 63)] = zr[q];
        }
    }
    __syncthreads();

    int buf = 0;
    for (int m = m_begin; m < m_end - 1; ++m) {
        const int64_t nb = (int64_t)(m + 1) * MC * 8;  // byte offset of the next macro-chunk
        const int64_t nbw = (int64_t)(m + 1) * MC * WB;
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int q = 0; q < ZPT; ++q)
                zr[q] = *(const double2 *)(zp + nb + s * (SYMM_SUB * 8) + q * zstep);
            const double *zrow = &zs[buf][0][s * SYMM_SUB + 2 * lane];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const double2 zz = *(const double2 *)(zrow + k * MC);
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    acc[i][k] = fma((double)a[s][i].x, zz.x, acc[i][k]);
                    acc[i][k] = fma((double)a[s][i].y, zz.y, acc[i][k]);
                }
            }
            // refill this pipeline stage for the next macro-chunk
#pragma unroll
            for (int i = 0; i < RPW; ++i)
                a[s][i] = *(const wpair *)(rowp[i] + nbw + s * (SYMM_SUB * WB));
#pragma unroll
            for (int q = 0; q < ZPT; ++q)
                *(double2 *)&zs[buf ^ 1][(tid >> 6) + 4 * q][s * SYMM_SUB + 2 * (tid & 63)] = zr[q];
        }
        __syncthreads();
        buf ^= 1;
    }
    // ---- last macro-chunk: nothing left to prefetch
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const double *zrow = &zs[buf][0][s * SYMM_SUB + 2 * lane];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const double2 zz = *(const double2 *)(zrow + k * MC);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                acc[i][k] = fma((double)a[s][i].x, zz.x, acc[i][k]);
                acc[i][k] = fma((double)a[s][i].y, zz.y, acc[i][k]);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int k = 0; k < B; ++k) {
            double s = acc[i][k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[i][k] = s;
        }
    if (lane == 0) {
        double *yp = ypart + (int64_t)blockIdx.y * rows * B;
#pragma unroll
        for (int i = 0; i < RPW; ++i)
            if (r0 + i < rows) {
#pragma unroll
                for (int k = 0; k < B; ++k) yp[(int64_t)(r0 + i) * B + k] = acc[i][k];
            }
    }
}

// y[r][:] = dinv[row_begin + r] * sum_seg ypart[seg][r][:]   (fixed order)
__global__ void k_symm_finish(const double *__restrict__ ypart, int nseg, int rows, int b,
                              const double *__restrict__ dinv, int row_begin,
                              double *__restrict__ y) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * b) return;
    double s = 0.0;
    for (int g = 0; g < nseg; ++g) s += ypart[(int64_t)g * rows * b + idx];
    y[idx] = dinv[row_begin + idx / b] * s;
}
