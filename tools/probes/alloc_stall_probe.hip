// Probe (round 6, not part of the product): does hipMalloc / hipFree / hipMallocAsync wait for a kernel that is
// running on another stream?  Build: hipcc --offload-arch=gfx950 -O2 -o alloc_stall_probe alloc_stall_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>

__global__ void k_spin(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                       \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int *sink;
    CK(hipMalloc(&sink, 4));
    const long long second = 100000000LL;  // wall_clock64 ticks at 100 MHz
    for (int busy = 0; busy < 2; ++busy) {
        for (size_t gb : {1, 8}) {
            void *p = nullptr, *q = nullptr;
            if (busy) k_spin<<<256, 64, 0, s1>>>(2 * second, sink);
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            double t0 = now();
            CK(hipMalloc(&p, gb << 30));
            double t1 = now();
            CK(hipMemsetAsync(p, 0, gb << 30, s2));
            CK(hipStreamSynchronize(s2));
            double t2 = now();
            CK(hipFree(p));
            double t3 = now();
            CK(hipMalloc(&p, gb << 30));
            double t4 = now();
            CK(hipMallocAsync(&q, gb << 30, s2));
            CK(hipStreamSynchronize(s2));
            double t5 = now();
            CK(hipFreeAsync(q, s2));
            CK(hipStreamSynchronize(s2));
            double t6 = now();
            CK(hipMallocAsync(&q, gb << 30, s2));
            CK(hipStreamSynchronize(s2));
            double t7 = now();
            printf("kernel running on another stream: %d, %zu GB: hipMalloc %.3f s, memset %.3f s, hipFree %.3f s, hipMalloc again %.3f s, "
                   "hipMallocAsync+sync %.3f s, hipFreeAsync+sync %.3f s, hipMallocAsync again %.3f s\n",
                   busy, gb, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6);
            CK(hipStreamSynchronize(s1));
            CK(hipFree(p));
            CK(hipFreeAsync(q, s2));
            CK(hipStreamSynchronize(s2));
        }
    }
    // a second thread that frees while the kernel runs, and this thread allocating at the same time
    {
        void *a = nullptr;
        CK(hipMalloc(&a, (size_t)4 << 30));
        k_spin<<<256, 64, 0, s1>>>(2 * second, sink);
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        double f0 = 0, f1 = 0;
        std::thread t([&] {
            f0 = now();
            hipFree(a);
            f1 = now();
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        void *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, (size_t)1 << 30));
        double t1 = now();
        t.join();
        printf("kernel running; another thread's hipFree of 4 GB took %.3f s; this thread's hipMalloc of 1 GB meanwhile %.3f s\n",
               f1 - f0, t1 - t0);
        CK(hipStreamSynchronize(s1));
        CK(hipFree(p));
    }
    return 0;
}
