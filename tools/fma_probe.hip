// Microbenchmark: sustained fp64 / fp32 FMA rate and LDS read rate on this box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T, int NACC>
__global__ __launch_bounds__(256) void k_fma(T *out, int iters, T x, T y) {
    T acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (T)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = acc[i] * x + y;
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_lds(double *out, int iters) {
    __shared__ double buf[8][1024];
    for (int i = threadIdx.x; i < 8 * 1024; i += 256) ((double *)buf)[i] = i;
    __syncthreads();
    double2 s = {0, 0};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double2 v = *(const double2 *)&buf[k][(2 * lane + 128 * (it & 3)) & 1023];
            s.x += v.x;
            s.y += v.y;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

int main() {
    double *out;
    CK(hipMalloc(&out, 8192 * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        const int iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            k_fma<double, 16><<<blocks, 256>>>(out, iters, 1.0000001, 1e-9);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("f64 fma: blocks %4d: %.3f ms -> %.2f TFLOP/s\n", blocks, ms,
                            2.0 * blocks * 256.0 * iters * 16 / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            k_fma<float, 16><<<blocks, 256>>>((float *)out, iters, 1.0000001f, 1e-9f);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("f32 fma: blocks %4d: %.3f ms -> %.2f TFLOP/s\n", blocks, ms,
                            2.0 * blocks * 256.0 * iters * 16 / ms / 1e9);
        }
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            k_lds<<<blocks, 256>>>(out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("lds b128: blocks %4d: %.3f ms -> %.1f TB/s\n", blocks, ms,
                            16.0 * blocks * 256.0 * iters * 8 / ms / 1e9);
        }
    }
    return 0;
}
