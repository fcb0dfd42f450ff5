// Probe (round 6, not part of the product): 80 GB of contiguous device address space backed by physical pieces
// (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess) beside one hipMalloc of 80 GB, in a
// process that holds other allocations already.   alloc_vmm_probe [GB per piece]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x)                                                \
    do {                                                     \
        hipError_t e_ = (x);                                 \
        if (e_ != hipSuccess) {                              \
            printf("%s -> %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                        \
        }                                                    \
    } while (0)

__global__ void k_fill(double *p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
__global__ void k_sum(const double *p, size_t n, double *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += stride) s += p[i];
    atomicAdd(out, s);
}

int main(int argc, char **argv) {
    const size_t piece_gb = argc > 1 ? (size_t)atoi(argv[1]) : 2;
    const size_t total = (size_t)80 << 30, piece = piece_gb << 30;
    void *held = nullptr;
    CK(hipMalloc(&held, (size_t)30 << 30));
    CK(hipMemset(held, 1, (size_t)30 << 30));
    CK(hipDeviceSynchronize());
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu bytes\n", gran);
    double t0 = now();
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    double t1 = now();
    std::vector<hipMemGenericAllocationHandle_t> handles;
    for (size_t off = 0; off < total; off += piece) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, piece, &prop, 0));
        CK(hipMemMap((char *)va + off, piece, 0, h, 0));
        handles.push_back(h);
    }
    double t2 = now();
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    double t3 = now();
    double *sum = nullptr;
    CK(hipMalloc(&sum, 8));
    CK(hipMemset(sum, 0, 8));
    k_fill<<<4096, 256>>>((double *)va, total / 8, 0.5);
    CK(hipDeviceSynchronize());
    double t4 = now();
    k_sum<<<4096, 256>>>((const double *)va, total / 8, sum);
    double h_sum = 0;
    CK(hipMemcpy(&h_sum, sum, 8, hipMemcpyDeviceToHost));
    double t5 = now();
    printf("30 GB held; 80 GB in pieces of %zu GB: reserve %.3f s, create+map %.3f s, set access %.3f s (total %.3f s); fill kernel %.3f s, "
           "sum kernel %.3f s (%.1f GB/s), sum %s\n",
           piece_gb, t1 - t0, t2 - t1, t3 - t2, t3 - t0, t4 - t3, t5 - t4, total / (t5 - t4) / 1e9,
           h_sum == 0.5 * (double)(total / 8) ? "exact" : "WRONG");
    double t6 = now();
    CK(hipMemUnmap(va, total));
    for (auto h : handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(va, total));
    double t7 = now();
    printf("unmap + release + free of the range: %.3f s\n", t7 - t6);
    // and the plain call, for comparison, in the same process
    void *p = nullptr;
    double t8 = now();
    CK(hipMalloc(&p, total));
    double t9 = now();
    k_sum<<<4096, 256>>>((const double *)p, total / 8, sum);
    CK(hipDeviceSynchronize());
    double t10 = now();
    printf("hipMalloc of 80 GB afterwards: %.3f s; sum kernel over it %.3f s (%.1f GB/s)\n", t9 - t8, t10 - t9, total / (t10 - t9) / 1e9);
    return 0;
}
