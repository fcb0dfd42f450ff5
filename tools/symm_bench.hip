// Tuning harness for k_symm (not part of the library): times (B, RPW, segments)
// variants on a random N x N matrix and checks them against a host reference.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../spectralclustersupertree_amd/csrc/scs_symm.h"

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e = (x);                                                         \
        if (e != hipSuccess) {                                                      \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

__global__ void k_read_probe(const double2 *__restrict__ p, size_t n2, double *out) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = p[i];
        s += v.x + v.y;
    }
    if (s == 123.456) out[0] = s;
}

template <int B, int RPW, int S, int MINW = 2>
double run(const double *w, int64_t ld, int n, const double *z, double *ypart, double *y,
           const double *dinv, int nseg, int reps) {
    const int n_chunks = (int)(ld / (S * SYMM_SUB));
    const int cps = (n_chunks + nseg - 1) / nseg;
    dim3 grid((n + 4 * RPW - 1) / (4 * RPW), nseg);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) {
        k_symm<B, RPW, S, MINW><<<grid, 256>>>(w, ld, n, z, ypart, cps);
        k_symm_finish<<<(n * B + 255) / 256, 256>>>(ypart, nseg, n, B, dinv, 0, y);
    }
    CK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        k_symm<B, RPW, S, MINW><<<grid, 256>>>(w, ld, n, z, ypart, cps);
        CK(hipEventRecord(e1));
        k_symm_finish<<<(n * B + 255) / 256, 256>>>(ypart, nseg, n, B, dinv, 0, y);
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        tot += ms;
    }
    CK(hipDeviceSynchronize());
    const double bytes = 8.0 * n * (double)n;
    printf("B=%2d RPW=%d S=%d MINW=%d seg=%2d grid=%5d x %2d : avg %.4f ms (%.0f GB/s)  best %.4f ms (%.0f GB/s)\n",
           B, RPW, S, MINW, nseg, grid.x, grid.y, tot / reps, bytes / (tot / reps) / 1e6, best,
           bytes / best / 1e6);
    return tot / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10000;
    const int64_t ld = (n + 511) / 512 * 512;
    std::vector<double> hw((size_t)n * ld), hz((size_t)ld * 16), hd(n, 1.0);
    srand(1);
    for (auto &v : hw) v = (double)rand() / RAND_MAX;
    for (int r = 0; r < n; ++r)
        for (int64_t c = n; c < ld; ++c) hw[(size_t)r * ld + c] = 0.0;
    for (auto &v : hz) v = (double)rand() / RAND_MAX - 0.5;
    double *w, *z, *yp, *y, *dinv;
    CK(hipMalloc(&w, hw.size() * 8));
    CK(hipMalloc(&z, hz.size() * 8));
    CK(hipMalloc(&yp, (size_t)96 * n * 16 * 8));
    CK(hipMalloc(&y, (size_t)n * 16 * 8));
    CK(hipMalloc(&dinv, (size_t)n * 8));
    CK(hipMemcpy(w, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(z, hz.data(), hz.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dinv, hd.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    const int reps = 20;
    // correctness of one variant against the host (B = 8 uses the first 8 values of each z row)
    {
        run<8, 4, 2, 2>(w, ld, n, z, yp, y, dinv, 3, 1);
        std::vector<double> hy((size_t)n * 8);
        CK(hipMemcpy(hy.data(), y, hy.size() * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int r = 0; r < n; r += n / 7 + 1)
            for (int k = 0; k < 8; ++k) {
                double s = 0;
                for (int j = 0; j < n; ++j) s += hw[(size_t)r * ld + j] * hz[(size_t)k * ld + j];
                worst = fmax(worst, fabs(s - hy[(size_t)r * 8 + k]));
            }
        printf("check: max abs err %.3e\n", worst);
    }
    // read-only streaming probe: the practical ceiling for this box
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const size_t n2 = (size_t)n * ld / 2;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            k_read_probe<<<2048, 256>>>((const double2 *)w, n2, y);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("read probe: %.4f ms (%.0f GB/s)\n", ms, 8.0 * n * (double)ld / ms / 1e6);
        }
    }
    for (int seg : {1, 2, 4}) {
        run<8, 4, 1, 2>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<8, 4, 2, 2>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<8, 4, 4, 2>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<8, 4, 2, 3>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<8, 2, 4, 3>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<8, 2, 4, 4>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<8, 8, 2, 1>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<4, 4, 4, 3>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<4, 8, 2, 2>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<16, 2, 2, 2>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<16, 4, 2, 1>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<12, 4, 2, 1>(w, ld, n, z, yp, y, dinv, seg, reps);
        run<12, 2, 4, 2>(w, ld, n, z, yp, y, dinv, seg, reps);
    }
    return 0;
}
