// Shader clock / dependent-op latency probe: one wave, dependent FMA chain, LDS round trips, barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(double *out, int reps) {
    __shared__ double lds[256];
    const int tid = threadIdx.x;
    lds[tid] = tid;
    __syncthreads();
    long long w0 = wall_clock64(), c0 = clock64();
    double x = 1.0 + tid;
    for (int i = 0; i < reps; ++i) x = fma(x, 1.0000001, 1e-9);
    long long w1 = wall_clock64(), c1 = clock64();
    // dependent LDS round trips
    int idx = tid;
    for (int i = 0; i < reps; ++i) idx = (int)lds[idx & 255] & 255;
    long long w2 = wall_clock64(), c2 = clock64();
    for (int i = 0; i < reps; ++i) __syncthreads();
    long long w3 = wall_clock64(), c3 = clock64();
    // dependent transcendental chain
    double y = 1.5 + tid;
    for (int i = 0; i < reps; ++i) y = __builtin_amdgcn_rsq(y) + 1.0;
    long long w4 = wall_clock64(), c4 = clock64();
    if (tid == 0) {
        out[0] = (double)(w1 - w0); out[1] = (double)(c1 - c0);
        out[2] = (double)(w2 - w1); out[3] = (double)(c2 - c1);
        out[4] = (double)(w3 - w2); out[5] = (double)(c3 - c2);
        out[6] = (double)(w4 - w3); out[7] = (double)(c4 - c3);
        out[8] = x + idx + y;
    }
}
int main() {
    double *d; hipMalloc(&d, 128);
    for (int threads : {64, 256}) for (int it = 0; it < 2; ++it) {
        const int reps = 20000;
        k_probe<<<1, threads>>>(d, reps);
        double h[9]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("threads %d: wall ticks (100 MHz) vs clock64 per op: fma %.2f ns %.2f clk | lds %.2f ns %.2f clk | barrier %.2f ns %.2f clk | rsq+add %.2f ns %.2f clk\n", threads,
               h[0] * 10 / reps, h[1] / reps, h[2] * 10 / reps, h[3] / reps, h[4] * 10 / reps, h[5] / reps, h[6] * 10 / reps, h[7] / reps);
    }
    return 0;
}
