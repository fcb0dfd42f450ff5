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
//        tools/cells_probe.hip -o tools/cells_probe      Run: tools/cells_probe [trees]
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
    run_d<4>(nt, 256 * 3 * 4);
    run_d<8>(nt, 256 * 6);
    run_d<12>(nt, 256 * 4);
    return 0;
}
