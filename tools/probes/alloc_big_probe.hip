// Probe (round 6, not part of the product): what does hipMalloc of 80 GB cost, fresh and after the process has
// used, kept or released other memory?   hipcc --offload-arch=gfx950 -O2 -o alloc_big_probe alloc_big_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

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

static int timed_malloc(const char *what, size_t gb, void **out) {
    double t0 = now();
    CK(hipMalloc(out, gb << 30));
    double t1 = now();
    CK(hipMemset(*out, 1, gb << 30));
    CK(hipDeviceSynchronize());
    double t2 = now();
    printf("%-64s hipMalloc %3zu GB %.3f s, first touch (memset) %.3f s\n", what, gb, t1 - t0, t2 - t1);
    return 0;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    void *a = nullptr, *b = nullptr, *c = nullptr;
    if (mode == 0) {
        if (timed_malloc("fresh process", 80, &a)) return 1;
        CK(hipFree(a));
        if (timed_malloc("again after hipFree of the same 80 GB", 80, &a)) return 1;
    } else if (mode == 1) {
        if (timed_malloc("fresh process", 24, &a)) return 1;
        if (timed_malloc("24 GB held", 20, &b)) return 1;
        if (timed_malloc("44 GB held", 80, &c)) return 1;
    } else if (mode == 2) {
        if (timed_malloc("fresh process", 30, &a)) return 1;
        CK(hipFree(a));
        if (timed_malloc("30 GB used and released", 80, &c)) return 1;
    } else {
        void *p[40];
        double t0 = now();
        for (int i = 0; i < 40; ++i) CK(hipMalloc(&p[i], (size_t)2 << 30));
        double t1 = now();
        printf("40 x hipMalloc of 2 GB: %.3f s\n", t1 - t0);
        if (timed_malloc("80 GB held in 40 pieces", 80, &c)) return 1;
    }
    return 0;
}
