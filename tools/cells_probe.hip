// Measurement, not product code: cycles per (workgroup, tree) step of two inner-loop shapes of
// the PCG tile kernel on one MI355X, with the table expansion's stores and the per-tree barrier
// standing in for the rest of the step.
//   L  lanes = columns (k_accumulate_mono): per cell one ds_read_b64 of a table row picked per
//      lane (2-way bank conflicts), v_min_f64, v_add_f64.
//   D  lanes = rows: per column one conflict-free ds_read_b64 (64 consecutive doubles),
//      v_add_u32 for the address, v_min_f64 with a SCALAR operand, v_add_f64; the per-column
//      scalars are stored by the wave itself with vector stores a step earlier and come back
//      through s_load (s_dcache_inv) -- the probe also CHECKS that hand-off (sums against the host).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I spectralclustersupertree_amd/csrc
//        -I tools tools/cells_probe.hip -o tools/cells_probe      Run: tools/cells_probe [trees] [x: stop before D]
//   H / LV (round 5): the rank-halved table with VGPR-indexed accumulators, and the product's loop with
//      fewer instructions per cell -- profiles/r05_cells_probe_rank_halved.txt
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "scs_cells_asm.h"
#include "cells_probe_asm.h"

constexpr int LD = 65;
constexpr int TB = 64 * LD;  // doubles per table buffer

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ double table_value(int r, int i) { return 1.0 + (double)((r * 64 + i) % 977) * 0.03125; }
__device__ __forceinline__ int pick_row(int c, int t) { return (c * 7 + t * 13 + (c >> 3)) & 63; }

// ---- L: WAVES waves per workgroup, thread = column
template <int WAVES, int NBUF>
__global__ __launch_bounds__(WAVES * 64) void k_probe_l(int nt, double *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) double s_tb[];  // [NBUF][TB]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < NBUF * TB; e += WAVES * 64) {
        const int r = (e % TB) / LD, i = (e % TB) % LD;
        s_tb[e] = i < 64 ? table_value(r, i) : 0.0;
    }
    __syncthreads();
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    const double vn = 1e300;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int t = 0; t < nt; ++t) {
        const int nb = pick_row(tid, t);
        const unsigned addr =
            (unsigned)(size_t)(__attribute__((address_space(3))) double *)&s_tb[(NBUF == 2 ? (t & 1) : 0) * TB + nb * LD];
        // expansion stand-in: 64 / WAVES steps of two stores (the values already there) into the other
        // buffer -- or, one buffer, into the same one with a second barrier as in k_accumulate_mono
        double *o = s_tb + (NBUF == 2 ? ((t + 1) & 1) : 0) * TB;
        const int wave = tid >> 6;
#pragma unroll
        for (int j = 0; j < 64 / WAVES; ++j) {
            const int b = wave * (64 / WAVES) + j;
            o[lane * LD + b] = table_value(lane, b);
            o[b * LD + lane] = table_value(b, lane);
        }
        if (NBUF == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        double tmp[SCS_CELLS_DEPTH];
        SCS_CELLS_ASM(acc, tmp, addr, vn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- D: WAVES waves per workgroup, lane = row, a wave walks 64 columns
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_probe_d(int nt, double *out, unsigned long long *cyc,
                                                         unsigned char *handoff) {
    extern __shared__ __attribute__((aligned(16))) double s_tb[];  // [2][TB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 2 * TB; e += WAVES * 64) {
        const int r = (e % TB) / LD, i = (e % TB) % LD;
        s_tb[e] = i < 64 ? table_value(r, i) : 0.0;
    }
    __syncthreads();
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    // this wave's hand-off region: 64 values (512 bytes) then 64 offsets (256 bytes)
    unsigned char *mine = handoff + ((size_t)blockIdx.x * WAVES + wave) * 1024;
    const unsigned lane8 = (unsigned)lane * 8u;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int t = 0; t < nt; ++t) {
        // "column phase": lane = column c of this wave; its table row and its own value
        const int c = wave * 64 + lane;
        const int nb = pick_row(c, t);
        const double vn = (c + t) % 5 == 0 ? 2.0 + (double)((c + t) % 29) : 1e300;
        const unsigned tbase = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&s_tb[(t & 1) * TB];
        ((double *)mine)[lane] = vn;
        ((unsigned *)(mine + 512))[lane] = tbase + (unsigned)nb * LD * 8u;
        double *o = s_tb + ((t + 1) & 1) * TB;
        // first two of the wave's expansion steps here, six more inside the statement
        const unsigned w1 = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&o[lane * LD + wave * 5];
        const unsigned w2 = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&o[(wave * 5) * LD + lane];
        const double wv = table_value(lane, wave * 5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached L2
        double tmp[8];
        unsigned adr[8];
        const unsigned char *base = mine;
        PROBE_D_ASM(acc, tmp, adr, lane8, base, w1, w2, wv);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    // acc[c] of lane i = sum over trees of min(T[nb(c, t)][i], vn(c, t))
#pragma unroll
    for (int cc = 0; cc < 64; ++cc) out[((size_t)blockIdx.x * WAVES * 64 + wave * 64 + cc) * 64 + lane] = acc[cc];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}


// ---- H (round 5): the rank-halved cell loop.  CONS consumer waves walk the 64 RANKS of the tile's
// rows (ranks 0-31 against half-table A -- 33 rows x 32 columns, stride 33 doubles -- ranks 32-63
// against half-table B: conflict-free reads), the cell goes into the accumulator of the rank's
// ORIGINAL row through VGPR index mode (M0 from wave-uniform 16-bit entries, new per tree);
// WAVES - CONS waves only join the barrier (the producers' places).  Checked against the host.
constexpr int HLD = 33;
constexpr int HA = 33 * HLD, HB = 32 * HLD, HT = HA + HB;  // doubles per table buffer
__host__ __device__ inline double h_table(int half, int row, int col) {
    return 1.0 + (double)((half * 1201 + (row + col) * 37 + row * col * 5) % 977) * 0.03125;  // symmetric
}
__host__ __device__ inline int h_row_a(int c, int t) { return (c * 7 + t * 13 + (c >> 3)) % 33; }
__host__ __device__ inline int h_row_b(int c, int t) { return (c * 5 + t * 11 + (c >> 2)) & 31; }
__host__ __device__ inline double h_cap_a(int c, int t) { return (c + t) % 5 == 0 ? 2.0 + (double)((c + t) % 29) : 1e300; }
__host__ __device__ inline double h_cap_b(int c, int t) { return (c + 2 * t) % 3 == 0 ? 3.0 + (double)((c + t) % 17) : 1e300; }

template <int WAVES, int CONS, int V>
__global__ __launch_bounds__(WAVES * 64) void k_probe_h(int nt, const unsigned *perm, double *out,
                                                         unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) double s_tb[];  // [2][HT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 2 * HT; e += WAVES * 64) {
        const int f = e % HT;
        const int half = f >= HA, g = half ? f - HA : f;
        s_tb[e] = h_table(half, g / HLD, g % HLD);
    }
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (wave >= CONS) {
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    typedef const __attribute__((address_space(4))) int *cint;
    for (int t = 0; t < nt; ++t) {
        const int c = tid;
        const double *tb = s_tb + (t & 1) * HT;
        const unsigned addrA = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)&tb[h_row_a(c, t) * HLD];
        const unsigned addrB = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)&tb[HA + h_row_b(c, t) * HLD];
        const double capA = h_cap_a(c, t), capB = h_cap_b(c, t);
        cint pp = (cint)(size_t)(perm + (size_t)t * 32);
        int P[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) P[j] = pp[j];
        // expansion stand-in: four steps of two stores into the other buffer (the values already there)
        double *o = s_tb + ((t + 1) & 1) * HT;
        const int b = (wave * 4) & 31;
        const unsigned w1 = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&o[(lane & 31) * HLD + b + (lane >> 5) * HA];
        const unsigned w2 = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&o[b * HLD + (lane & 31) + (lane >> 5) * HA];
        const double wv = h_table(lane >> 5, lane & 31, b);
        double tmp[16];
        int st;
        if constexpr (V == 0) PROBE_H_ASM_V0(acc, tmp, st, addrA, addrB, capA, capB, P, w1, w2, wv);
        else if constexpr (V == 1) PROBE_H_ASM_V1(acc, tmp, st, addrA, addrB, capA, capB, P, w1, w2, wv);
        else if constexpr (V == 2) PROBE_H_ASM_V2(acc, tmp, st, addrA, addrB, capA, capB, P, w1, w2, wv);
        else if constexpr (V == 3) PROBE_H_ASM_V3(acc, tmp, st, addrA, addrB, capA, capB, P, w1, w2, wv);
        else if constexpr (V == 4) PROBE_H_ASM_V4(acc, tmp, st, addrA, addrB, capA, capB, P, w1, w2, wv);
        else if constexpr (V == 5) PROBE_H_ASM_V5(acc, tmp, st, addrA, addrB, capA, capB, P, w1, w2, wv);
        else PROBE_H_ASM_V6(acc, tmp, st, addrA, addrB, capA, capB, P, w1, w2, wv);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) {
        const int which = blockIdx.x == 0 ? 0 : 1;
#pragma unroll
        for (int i = 0; i < 64; ++i) out[((size_t)which * CONS * 64 + tid) * 64 + i] = acc[i];
    }
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WAVES, int CONS, int V>
static void run_h(int nt, int blocks) {
    // per tree a permutation rank -> original row, as 16-bit entries 0x9000 | 2 orig
    std::vector<unsigned> perm((size_t)nt * 32);
    std::vector<int> orig((size_t)nt * 64);
    for (int t = 0; t < nt; ++t) {
        int p[64];
        for (int r = 0; r < 64; ++r) p[r] = r;
        unsigned s = 12345u + 977u * t;
        for (int r = 63; r > 0; --r) {
            s = s * 1664525u + 1013904223u;
            const int j = (s >> 8) % (r + 1);
            const int x = p[r];
            p[r] = p[j];
            p[j] = x;
        }
        for (int r = 0; r < 64; ++r) {
            orig[(size_t)t * 64 + r] = p[r];
            const unsigned e = 0x9000u | (unsigned)(2 * p[r]);
            if (r & 1) perm[(size_t)t * 32 + (r >> 1)] |= e << 16;
            else perm[(size_t)t * 32 + (r >> 1)] = e;
        }
    }
    unsigned *d_perm;
    double *d_out;
    unsigned long long *d_cyc;
    const size_t n_out = (size_t)2 * CONS * 64 * 64;
    CK(hipMalloc(&d_perm, perm.size() * 4));
    CK(hipMemcpy(d_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, n_out * 8));
    CK(hipMalloc(&d_cyc, blocks * 8));
    CK(hipMemset(d_cyc, 0, blocks * 8));
    const size_t lds = 2 * HT * 8 + 6 * 2800 + 24576;  // + records and hand-off arrays of a real kernel
    CK(hipFuncSetAttribute((const void *)k_probe_h<WAVES, CONS, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    k_probe_h<WAVES, CONS, V><<<blocks, WAVES * 64, lds>>>(nt, d_perm, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_probe_h<WAVES, CONS, V><<<blocks, WAVES * 64, lds>>>(nt, d_perm, d_out, d_cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<double> out(n_out);
    std::vector<unsigned long long> cyc(blocks);
    CK(hipMemcpy(out.data(), d_out, n_out * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int which = 0; which < 2; ++which)
        for (int c = 0; c < CONS * 64; ++c) {
            double want[64];
            for (int i = 0; i < 64; ++i) want[i] = 0.0;
            for (int t = 0; t < nt; ++t)
                for (int r = 0; r < 64; ++r) {
                    const int half = r >= 32;
                    const double e = half ? h_table(1, h_row_b(c, t), r - 32) : h_table(0, h_row_a(c, t), r);
                    want[orig[(size_t)t * 64 + r]] += std::fmin(e, half ? h_cap_b(c, t) : h_cap_a(c, t));
                }
            for (int i = 0; i < 64; ++i)
                if (out[((size_t)which * CONS * 64 + c) * 64 + i] != want[i]) {
                    if (bad < 3 && V == 0) printf("   c %d i %d got %.6f want %.6f\n", c, i, out[((size_t)which * CONS * 64 + c) * 64 + i], want[i]);
                    ++bad;
                }
        }
    double mean = 0;
    for (auto v : cyc) mean += (double)v;
    mean /= blocks;
    const double cells = (double)blocks * CONS * 64 * 64 * nt;
    printf("H%d waves/WG %2d (%d consumers)  WGs %5d  trees %4d: %8.3f ms  %.3e cell-trees/s  s_memtime ticks per step %.0f  "
           "check: %zu cells wrong of %d\n",
           V, WAVES, CONS, blocks, nt, ms, cells / (ms * 1e-3), mean / nt, bad, 2 * CONS * 64 * 64);
    hipFree(d_perm);
    hipFree(d_out);
    hipFree(d_cyc);
}

// ---- LV (round 5): the column-lane loop of k_probe_l<WAVES, 2> with fewer instructions per cell
// (PROBE_L_ASM_V*: shared waits, ds_read_b128 per two cells on an even row stride); sums checked.
template <int WAVES, int V, int LDV>
__global__ __launch_bounds__(WAVES * 64) void k_probe_lv(int nt, double *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) double s_tb[];  // [2][64 * LDV]
    constexpr int TBV = 64 * LDV;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int e = tid; e < 2 * TBV; e += WAVES * 64) {
        const int r = (e % TBV) / LDV, i = (e % TBV) % LDV;
        s_tb[e] = i < 64 ? table_value(r, i) : 0.0;
    }
    __syncthreads();
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    const double vn = 1e300;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int t = 0; t < nt; ++t) {
        const int nb = pick_row(tid, t);
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&s_tb[(t & 1) * TBV + nb * LDV];
        double *o = s_tb + ((t + 1) & 1) * TBV;
        const int wave = tid >> 6;
#pragma unroll
        for (int j = 0; j < 64 / WAVES; ++j) {
            const int b = wave * (64 / WAVES) + j;
            o[lane * LDV + b] = table_value(lane, b);
            o[b * LDV + lane] = table_value(b, lane);
        }
        double tmp[16];
        if constexpr (V == 0) PROBE_L_ASM_V0(acc, tmp, addr, vn);
        else if constexpr (V == 1) PROBE_L_ASM_V1(acc, tmp, addr, vn);
        else if constexpr (V == 2) PROBE_L_ASM_V2(acc, tmp, addr, vn);
        else if constexpr (V == 3) PROBE_L_ASM_V3(acc, tmp, addr, vn);
        else if constexpr (V == 4) PROBE_L_ASM_V4(acc, tmp, addr, vn);
        else PROBE_L_ASM_V5(acc, tmp, addr, vn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

static double host_table_l(int r, int i) { return 1.0 + (double)((r * 64 + i) % 977) * 0.03125; }
static int host_row_l(int c, int t) { return (c * 7 + t * 13 + (c >> 3)) & 63; }

template <int WAVES, int V, int LDV>
static void run_lv(int nt, int blocks) {
    double *d_out;
    unsigned long long *d_cyc;
    const size_t n_out = (size_t)blocks * WAVES * 64;
    CK(hipMalloc(&d_out, n_out * 8));
    CK(hipMalloc(&d_cyc, blocks * 8));
    const size_t lds = 2 * 64 * LDV * 8 + 5504;
    CK(hipFuncSetAttribute((const void *)k_probe_lv<WAVES, V, LDV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    k_probe_lv<WAVES, V, LDV><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_probe_lv<WAVES, V, LDV><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> cyc(blocks);
    std::vector<double> out(n_out);
    CK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.data(), d_out, n_out * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int c = 0; c < WAVES * 64; ++c) {
        double acc[64];
        for (int i = 0; i < 64; ++i) acc[i] = 0.0;
        for (int t = 0; t < nt; ++t)
            for (int i = 0; i < 64; ++i) acc[i] += host_table_l(host_row_l(c, t), i);
        double s = 0.0;
        for (int i = 0; i < 64; ++i) s += acc[i];
        if (out[c] != s) ++bad;
    }
    double mean = 0;
    for (auto v : cyc) mean += (double)v;
    mean /= blocks;
    const double cells = (double)blocks * WAVES * 64 * 64 * nt;
    printf("LV%d waves/WG %2d  row stride %d  WGs %5d  trees %4d: %8.3f ms  %.3e cell-trees/s  s_memtime ticks per step %.0f  check: %zu of %d sums wrong\n",
           V, WAVES, LDV, blocks, nt, ms, cells / (ms * 1e-3), mean / nt, bad, WAVES * 64);
    hipFree(d_out);
    hipFree(d_cyc);
}

// ---- L32 (round 5): the product's loop with HALF a row block per wave -- 32 accumulators (64 registers)
// instead of 64, so that four or five waves fit a SIMD where three do now; two waves share a column's
// (table row, own value) pair.  WAVES waves per workgroup (all consumers), WGS workgroups per CU.
template <int WAVES, int V, int LDV, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_probe_l32(int nt, double *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) double s_tb[];  // [2][64 * LDV]
    constexpr int TBV = 64 * LDV;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 2 * TBV; e += WAVES * 64) {
        const int r = (e % TBV) / LDV, i = (e % TBV) % LDV;
        s_tb[e] = i < 64 ? table_value(r, i) : 0.0;
    }
    __syncthreads();
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    const double vn = 1e300;
    const int half = wave & 1, cgrp = wave >> 1;
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int t = 0; t < nt; ++t) {
        const int nb = pick_row(cgrp * 64 + lane, t);
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&s_tb[(t & 1) * TBV + nb * LDV + half * 32];
        double *o = s_tb + ((t + 1) & 1) * TBV;
#pragma unroll
        for (int j = 0; j < (64 + WAVES - 1) / WAVES; ++j) {
            const int b = (wave * ((64 + WAVES - 1) / WAVES) + j) & 63;
            o[lane * LDV + b] = table_value(lane, b);
            o[b * LDV + lane] = table_value(b, lane);
        }
        double tmp[16];
        if constexpr (V == 6) PROBE_L_ASM_V6(acc, tmp, addr, vn);
        else PROBE_L_ASM_V7(acc, tmp, addr, vn);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WAVES, int V, int LDV, int MINW>
static void run_l32(int nt, int blocks) {
    double *d_out;
    unsigned long long *d_cyc;
    const size_t n_out = (size_t)blocks * WAVES * 64;
    CK(hipMalloc(&d_out, n_out * 8));
    CK(hipMalloc(&d_cyc, blocks * 8));
    const size_t lds = 2 * 64 * LDV * 8 + 5504;
    CK(hipFuncSetAttribute((const void *)k_probe_l32<WAVES, V, LDV, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    k_probe_l32<WAVES, V, LDV, MINW><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_probe_l32<WAVES, V, LDV, MINW><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> cyc(blocks);
    std::vector<double> out(n_out);
    CK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.data(), d_out, n_out * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int w = 0; w < WAVES; ++w)
        for (int l = 0; l < 64; ++l) {
            double acc[32];
            for (int i = 0; i < 32; ++i) acc[i] = 0.0;
            for (int t = 0; t < nt; ++t)
                for (int i = 0; i < 32; ++i) acc[i] += host_table_l(host_row_l((w >> 1) * 64 + l, t), (w & 1) * 32 + i);
            double s = 0.0;
            for (int i = 0; i < 32; ++i) s += acc[i];
            if (out[w * 64 + l] != s) ++bad;
        }
    double mean = 0;
    for (auto v : cyc) mean += (double)v;
    mean /= blocks;
    const double cells = (double)blocks * WAVES * 64 * 32 * nt;
    printf("L32 V%d waves/WG %2d (%d WG/CU asked)  row stride %d  WGs %5d  trees %4d: %8.3f ms  %.3e cell-trees/s  s_memtime ticks per step %.0f  check: %zu of %d sums wrong\n",
           V, WAVES, MINW, LDV, blocks, nt, ms, cells / (ms * 1e-3), mean / nt, bad, WAVES * 64);
    hipFree(d_out);
    hipFree(d_cyc);
}

static double host_table(int r, int i) { return 1.0 + (double)((r * 64 + i) % 977) * 0.03125; }
static int host_row(int c, int t) { return (c * 7 + t * 13 + (c >> 3)) & 63; }

template <int WAVES>
static void run_d(int nt, int blocks) {
    double *d_out;
    unsigned long long *d_cyc;
    unsigned char *d_hand;
    const size_t n_out = (size_t)blocks * WAVES * 64 * 64;
    CK(hipMalloc(&d_out, n_out * 8));
    CK(hipMalloc(&d_cyc, blocks * 8));
    CK(hipMalloc(&d_hand, (size_t)blocks * WAVES * 1024));
    CK(hipMemset(d_hand, 0, (size_t)blocks * WAVES * 1024));
    const size_t lds = 2 * TB * 8;
    CK(hipFuncSetAttribute((const void *)k_probe_d<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    k_probe_d<WAVES><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc, d_hand);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_probe_d<WAVES><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc, d_hand);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<double> out(n_out);
    std::vector<unsigned long long> cyc(blocks);
    CK(hipMemcpy(out.data(), d_out, n_out * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
    // check workgroup 0 and the last one
    size_t bad = 0;
    for (int blk : {0, blocks - 1})
        for (int c = 0; c < WAVES * 64; ++c)
            for (int i = 0; i < 64; ++i) {
                double want = 0.0;
                for (int t = 0; t < nt; ++t) {
                    const double vn = (c + t) % 5 == 0 ? 2.0 + (double)((c + t) % 29) : 1e300;
                    want += std::fmin(host_table(host_row(c, t), i), vn);
                }
                if (out[((size_t)blk * WAVES * 64 + c) * 64 + i] != want) ++bad;
            }
    double mean = 0;
    for (auto v : cyc) mean += (double)v;
    mean /= blocks;
    const double cells = (double)blocks * WAVES * 64 * 64 * nt;
    printf("D  waves/WG %2d  WGs %5d  trees %4d: %8.3f ms  %.3e cell-trees/s  s_memtime ticks per step %.0f  "
           "hand-off check: %zu cells wrong of %d\n",
           WAVES, blocks, nt, ms, cells / (ms * 1e-3), mean / nt, bad, 2 * WAVES * 64 * 64);
    hipFree(d_out);
    hipFree(d_cyc);
    hipFree(d_hand);
}

template <int WAVES, int NBUF>
static void run_l(int nt, int blocks) {
    double *d_out;
    unsigned long long *d_cyc;
    const size_t n_out = (size_t)blocks * WAVES * 64;
    CK(hipMalloc(&d_out, n_out * 8));
    CK(hipMalloc(&d_cyc, blocks * 8));
    const size_t lds = NBUF * TB * 8 + 5504;  // + the two records of the real kernel
    CK(hipFuncSetAttribute((const void *)k_probe_l<WAVES, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    k_probe_l<WAVES, NBUF><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k_probe_l<WAVES, NBUF><<<blocks, WAVES * 64, lds>>>(nt, d_out, d_cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> cyc(blocks);
    CK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : cyc) mean += (double)v;
    mean /= blocks;
    const double cells = (double)blocks * WAVES * 64 * 64 * nt;
    printf("L  waves/WG %2d x %d table buffer(s)  WGs %5d  trees %4d: %8.3f ms  %.3e cell-trees/s  s_memtime ticks per step %.0f\n", WAVES,
           NBUF, blocks, nt, ms, cells / (ms * 1e-3), mean / nt);
    hipFree(d_out);
    hipFree(d_cyc);
}

int main(int argc, char **argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 256;
    // the same number of columns in every run: 256 CUs x 12 waves x 4 rounds
    run_l<4, 1>(nt, 256 * 3 * 4);
    run_l<4, 2>(nt, 256 * 3 * 4);
    run_l<12, 2>(nt, 256 * 4);
    run_l<8, 2>(nt, 256 * 6);
    run_l32<16, 6, 65, 1>(nt, 256 * 6);
    run_l32<16, 7, 66, 1>(nt, 256 * 6);
    run_l32<8, 6, 65, 2>(nt, 256 * 12);
    run_l32<8, 7, 66, 2>(nt, 256 * 12);
    run_l32<10, 6, 65, 2>(nt, 256 * 12);
    run_l32<12, 6, 65, 1>(nt, 256 * 8);
    run_lv<8, 0, 65>(nt, 256 * 6);
    run_lv<8, 1, 65>(nt, 256 * 6);
    run_lv<8, 2, 66>(nt, 256 * 6);
    run_lv<8, 3, 66>(nt, 256 * 6);
    run_h<8, 8, 0>(nt, 256 * 6);
    run_h<8, 8, 1>(nt, 256 * 6);
    run_h<8, 8, 2>(nt, 256 * 6);
    run_h<8, 8, 3>(nt, 256 * 6);
    run_h<8, 8, 4>(nt, 256 * 6);
    run_h<8, 8, 5>(nt, 256 * 6);
    run_h<8, 8, 6>(nt, 256 * 6);
    if (argc > 2) return 0;
    run_d<4>(nt, 256 * 3 * 4);
    run_d<8>(nt, 256 * 6);
    run_d<12>(nt, 256 * 4);
    return 0;
}
