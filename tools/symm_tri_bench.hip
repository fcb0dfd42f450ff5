// Measurement, not product code (round 5): what would a tile-major (column-panel) storage of W buy the
// symmetric SYMM at configs[3]'s size?  k_symm_tri<8, 2, 2, 4> and <4, 2, 4, 3> over the upper tiles of an
// n x n matrix, (a) row-major with the library's leading dimension -- a 128 x 256 tile is 128 pieces of
// 2 KB, 8 ld bytes apart -- and (b) column panels of 256 columns stored on their own (panel_stride): the
// tile is 256 KB in one piece.  Zeros for data (bandwidth does not care).
// Build: hipcc --offload-arch=gfx950 -O3 tools/symm_tri_bench.hip -o tools/symm_tri_bench    Run: tools/symm_tri_bench [n]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../spectralclustersupertree_amd/csrc/scs_symm_tri.h"

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

template <int B, int CT, int RPW, int D, typename WT = double>
static void run(const char *what, const WT *w, int64_t ld, int n, int64_t panel_stride, const double *z, int64_t ldz,
                const int2 *tiles, int n_tiles, double *pdir, double *ptr) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) k_symm_tri<B, CT, RPW, D, WT><<<n_tiles, 256>>>(w, ld, n, z, ldz, tiles, pdir, ptr, panel_stride);
    CK(hipDeviceSynchronize());
    float tot = 0, best = 1e30f;
    const int reps = 10;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        k_symm_tri<B, CT, RPW, D, WT><<<n_tiles, 256>>>(w, ld, n, z, ldz, tiles, pdir, ptr, panel_stride);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
        best = ms < best ? ms : best;
    }
    const double bytes = 1024.0 * CT * (double)n_tiles * TRI_TH;
    printf("k_symm_tri<%d, %d, %d, %d> %-28s n %6d: avg %8.3f ms (%6.0f GB/s)  best %8.3f ms (%6.0f GB/s)\n", B, CT, RPW, D, what,
           n, tot / reps, bytes / (tot / reps) / 1e6, best, bytes / best / 1e6);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 50000;
    const int64_t ld = ((int64_t)n + 511) / 512 * 512;
    const int tw = 256;
    const int n_rb = (n + TRI_TH - 1) / TRI_TH, n_ct = (n + tw - 1) / tw;
    const int64_t rows_alloc = (int64_t)n_rb * TRI_TH;
    const int64_t panel_stride = rows_alloc * tw;
    std::vector<int2> tiles;
    for (int i = 0; i < n_rb; ++i)
        for (int j = i * TRI_TH / tw; j < n_ct; ++j) tiles.push_back(make_int2(i, j));
    double *w, *z, *pdir, *ptr;
    int2 *d_tiles;
    const size_t w_bytes = (size_t)std::max<int64_t>(rows_alloc * ld, (int64_t)n_ct * panel_stride) * 8;
    CK(hipMalloc(&w, w_bytes));
    CK(hipMemset(w, 0, w_bytes));
    CK(hipMalloc(&z, (size_t)8 * ld * 8));
    CK(hipMemset(z, 0, (size_t)8 * ld * 8));
    CK(hipMalloc(&pdir, (size_t)n_ct * n * 8 * 8));
    CK(hipMalloc(&ptr, (size_t)n_rb * n * 8 * 8));
    CK(hipMalloc(&d_tiles, tiles.size() * sizeof(int2)));
    CK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));
    std::vector<int2> tiles512;
    const int n_ct512 = (n + 511) / 512;
    for (int i = 0; i < n_rb; ++i)
        for (int j = i * TRI_TH / 512; j < n_ct512; ++j) tiles512.push_back(make_int2(i, j));
    int2 *d_tiles512;
    const int n_tiles512 = (int)tiles512.size();
    CK(hipMalloc(&d_tiles512, tiles512.size() * sizeof(int2)));
    CK(hipMemcpy(d_tiles512, tiles512.data(), tiles512.size() * sizeof(int2), hipMemcpyHostToDevice));
    printf("%zu upper tiles of 128 x 256, %.2f GB streamed per application\n", tiles.size(), 8.0 * tiles.size() * TRI_TH * tw / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 2, 2, 4>("row-major (ld)", w, ld, n, 0, z, ld, d_tiles, (int)tiles.size(), pdir, ptr);
        run<8, 2, 2, 4>("column panels (tile-major)", w, ld, n, panel_stride, z, ld, d_tiles, (int)tiles.size(), pdir, ptr);
        run<4, 2, 4, 3>("row-major (ld)", w, ld, n, 0, z, ld, d_tiles, (int)tiles.size(), pdir, ptr);
        run<4, 2, 4, 3>("column panels (tile-major)", w, ld, n, panel_stride, z, ld, d_tiles, (int)tiles.size(), pdir, ptr);
        // round 5, mixed precision: the single-precision image of W (same tiles: 128 x 256, 1 KB row pieces)
        run<4, 1, 4, 3, float>("W32 row-major (ld)", (const float *)w, ld, n, 0, z, ld, d_tiles, (int)tiles.size(), pdir, ptr);
        run<4, 1, 4, 4, float>("W32 row-major (ld)", (const float *)w, ld, n, 0, z, ld, d_tiles, (int)tiles.size(), pdir, ptr);
        run<8, 1, 2, 4, float>("W32 row-major (ld)", (const float *)w, ld, n, 0, z, ld, d_tiles, (int)tiles.size(), pdir, ptr);
        // 512-column tiles of the image (2 KB row pieces): every second tile of the 256-column list
        run<4, 2, 4, 2, float>("W32, 128 x 512 tiles", (const float *)w, ld, n, 0, z, ld, d_tiles512, n_tiles512, pdir, ptr);
        run<4, 2, 2, 3, float>("W32, 128 x 512 tiles", (const float *)w, ld, n, 0, z, ld, d_tiles512, n_tiles512, pdir, ptr);
        run<4, 2, 2, 4, float>("W32, 128 x 512 tiles", (const float *)w, ld, n, 0, z, ld, d_tiles512, n_tiles512, pdir, ptr);
    }
    return 0;
}
