// Accuracy of the hardware v_rsq_f64 / v_rcp_f64 / v_sqrt_f64 seeds (gfx950).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *rs, double *rc, double *sq, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rs[i] = __builtin_amdgcn_rsq(x[i]);
    rc[i] = __builtin_amdgcn_rcp(x[i]);
    sq[i] = __builtin_amdgcn_sqrt(x[i]);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 40) - 20);
    }
    double *dx, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d1, d2, d3, n);
    hipMemcpy(a.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d2, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        e1 = std::fmax(e1, std::fabs(a[i] * std::sqrt(h[i]) - 1.0));
        e2 = std::fmax(e2, std::fabs(b[i] * h[i] - 1.0));
        e3 = std::fmax(e3, std::fabs(c[i] / std::sqrt(h[i]) - 1.0));
    }
    printf("max relative error: rsq %.3e  rcp %.3e  sqrt %.3e\n", e1, e2, e3);
    return 0;
}
