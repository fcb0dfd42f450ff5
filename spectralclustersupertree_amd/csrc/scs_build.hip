// Proper-cluster-graph build on gfx950: W (fp64, row block x V) from flattened trees.
//
// Replaces _proper_cluster_graph_edges / _dfs_pcg_weights and the dense fill of
// spectral_cluster_graph (reference: src/sc_supertree/scs.py:495-663, 246-250).
//
// Formulation (output-stationary, no global atomics, tree order preserved):
//   W[a][b] = sum_t  vw_t(LCA_t(a,b))      over trees where depth(LCA_t(a,b)) > 0
// with vw_t(node) = value(node) * w_t rounded once (scs.py:656).  A workgroup
// owns a 64 x 256 tile of W in registers (thread = column, 64 row accumulators)
// and walks the trees in order, so every cell sees its addends in the
// reference's order and the sum is bit-identical.
//
// LCA depth is an ultrametric.  For a tile's 64 rows Y and a column c in tree t:
//   d(i,c) = min( D[i][nb(c)], dn(c) ),  nb(c) = the row next to c in DFS order with the
//   deeper LCA, dn(c) = depth(LCA(nb(c), c)),
// where D is the 64 x 64 row-row LCA table of the tile in that tree.  D is expanded in LDS
// from a per-(row block, tree) record (rows sorted by DFS position + the 63 LCAs between
// neighbours); nb/dn cost ONE range-minimum query (two independent loads) per column on the
// tree's sparse table.  The inner loop is one ds_read_b64 per cell plus
//   * a v_min_f64 and a v_add_f64 when the value is monotone in the depth (scs_mono.h: the
//     table and the range-minimum tables then hold values), or
//   * a rank compare, a select and a v_add_f64 otherwise (scs_gen.h: (depth, value) pairs).

#include <algorithm>

#include "scs_internal.h"

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u32 DEPTH_INF = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------
// prep kernels
// ---------------------------------------------------------------------------

// ---- range-minimum tables ---------------------------------------------------
// Per tree with m gaps (gap p = the LCA of leaves p and p + 1 in DFS order): a plain sparse
// table, level k at [k m, (k + 1) m), entry p = the minimum over gaps [p, p + 2^k); a query is
// two independent loads.  Entries are the gap VALUES when the weighting is monotone in the
// depth (scs_mono.h), else (depth, value) pairs ordered by depth (scs_gen.h).
// (A blocked table -- in-block prefix/suffix minima plus a sparse table over block minima,
// 6.75 m entries -- was measured slower: twice the gathers per query cost more than the
// smaller footprint saved.)
// (So was a position-major table -- all levels of a position in one 128-byte line, so that a
// tile row's or a column's own entries share a line whatever the level: 7.9 against 7.2 ms at
// 10 000 leaves, 798 against 680 ms at 50 000, and the level kernels' scattered stores triple
// the preparation.)
// offsets (entries from the tree's table base) of the two loads of a query over gaps
// [a, b), 0 <= a < b <= m
__device__ __forceinline__ void rmq_offsets(int m, int a, int b, int (&o)[2]) {
    const int k = 31 - __clz(b - a);
    o[0] = k * m + a;
    o[1] = k * m + b - (1 << k);
}

template <typename K>
__device__ __forceinline__ K rmq_min(const K *__restrict__ base, int m, int a, int b) {
    int o[2];
    rmq_offsets(m, a, b, o);
    const K x = base[o[0]], y = base[o[1]];
    return x < y ? x : y;
}

// monotone builds: the range-minimum table is over the gap VALUES themselves (the value of
// the shallowest LCA of a range is the smallest value in it), level 0 = value * w per gap
__global__ void k_positions_values(const int64_t *__restrict__ tree_off,
                                   const int32_t *__restrict__ leaf_taxon,
                                   const int32_t *__restrict__ adj_depth,
                                   const double *__restrict__ adj_val,
                                   const double *__restrict__ tree_w, int t0,
                                   int32_t *__restrict__ pos, int64_t npad,
                                   const int64_t *__restrict__ st_off, double *__restrict__ stv) {
    const int tl = blockIdx.y;
    const int t = t0 + tl;
    const int64_t off = tree_off[t];
    const int n = (int)(tree_off[t + 1] - off);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    // (ids of table chunks that arrive behind the first tree batch are range-checked on the copy
    // stream and the verdict is read at the end of the build: an id nobody has vouched for yet
    // must not leave the tree's row of `pos` -- npad >= n_taxa)
    const unsigned tx = (unsigned)leaf_taxon[off + p];
    if (tx < (unsigned)npad) pos[(int64_t)tl * npad + tx] = p;
    if (p < n - 1)  // one rounded multiply, as the reference's `length * tree_weight`
        stv[st_off[tl] + p] = adj_depth[off + p] ? adj_val[off + p] * tree_w[t] : 0.0;
}

// level k >= 1 of every tree's sparse table; grid as k_positions
template <typename K>
__global__ void k_sparse_level(const int64_t *__restrict__ tree_off, int t0, int k,
                               const int64_t *__restrict__ st_off, K *__restrict__ st) {
    const int tl = blockIdx.y;
    const int m = (int)(tree_off[t0 + tl + 1] - tree_off[t0 + tl]) - 1;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < (1 << k) || p > m - (1 << k)) return;
    K *base = st + st_off[tl];
    const K a = base[(int64_t)(k - 1) * m + p], b = base[(int64_t)(k - 1) * m + p + (1 << (k - 1))];
    base[(int64_t)k * m + p] = a < b ? a : b;
}

// ALL levels k >= 1 of a tree's sparse table by ONE workgroup per tree (grid = trees in the batch):
// a launch per level is a dozen launches of a few microseconds of work each for trees of up to a
// few 10^4 leaves -- 0.56 ms of a 19 ms step at configs[2] (profiles/r04_bench_cfg2_kernel_stats.csv:
// 63 launches of k_sparse_level per pass).  Level k reads what the same workgroup wrote as level
// k - 1: __syncthreads() orders that (workgroup-scope release / acquire on global memory: the
// workgroup's waves share one L1).  MIN(a, b) is `a < b ? a : b` on values, gap_min on pairs.
template <typename K, typename MIN>
__global__ __launch_bounds__(1024) void k_sparse_levels_fused(const int64_t *__restrict__ tree_off, int t0,
                                                               const int64_t *__restrict__ st_off,
                                                               K *__restrict__ st, MIN pick) {
    const int tl = blockIdx.x;
    const int m = (int)(tree_off[t0 + tl + 1] - tree_off[t0 + tl]) - 1;
    K *base = st + st_off[tl];
    for (int k = 1; (1 << k) <= m; ++k) {
        const K *prev = base + (int64_t)(k - 1) * m;
        K *cur = base + (int64_t)k * m;
        const int half = 1 << (k - 1), last = m - (1 << k);
        for (int p = threadIdx.x; p <= last; p += 1024) cur[p] = pick(prev[p], prev[p + half]);
        __syncthreads();
    }
}
constexpr int SPARSE_FUSED_MAX_LEAVES = 32768;  // beyond: a launch per level has real work and more parallelism
struct pick_smaller_value {
    __device__ double operator()(double a, double b) const { return a < b ? a : b; }
};

// one v_min_f64 (the builtin fmin adds a canonicalising v_max_f64 in front of it);
// operands are finite non-negative values or +inf, so IEEE minNum semantics are moot
__device__ __forceinline__ double min_f64(double a, double b) {
#ifdef SCS_MIN_ASM
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return __builtin_fmin(a, b);
#endif
}

#include "scs_mono.h"  // monotone fast path: k_block_records_mono, k_accumulate_mono
#include "scs_mono_wide.h"  // the same walk for three column tiles of a row block per workgroup
#include "scs_gen.h"   // general path on the same tile structure: k_block_records_gen, k_accumulate_gen
struct pick_shallower_pair {
    __device__ gap_entry operator()(const gap_entry a, const gap_entry b) const { return gap_min(a, b); }
};

// ---------------------------------------------------------------------------
// comparison variant: input-stationary scatter with global fp64 atomics (SCS_BUILD_SCATTER)
// ---------------------------------------------------------------------------
// The formulation the north star's wording describes (reference loop
// structure: scs.py:644-658): walk the TREES, and for every pair of leaves of a tree that share
// a root side add value(LCA) * weight into W[a][b] and W[b][a] with atomicAdd.  A workgroup
// takes a 64 x 64 block of leaf-position pairs of one tree (coalesced loads of the two leaf
// ranges, the LCA value from the tree's sparse table), 16 pairs per thread.  The order of the
// adds is whatever the hardware makes it: results agree with the ordered build to rounding
// (<= 1e-12 relative), not bit for bit, and every pair update is a read-modify-write of 8 bytes
// at a random address of the 8 V^2-byte matrix (twice, for the mirror image).  Kept as the
// measured comparison only -- the product path is the output-stationary tile kernel.
__global__ __launch_bounds__(256) void k_scatter_atomic(const int64_t *__restrict__ tree_off,
                                                         const int32_t *__restrict__ leaf_taxon, int t0,
                                                         const int64_t *__restrict__ st_off,
                                                         const double *__restrict__ stv,
                                                         double *__restrict__ w, int64_t ld) {
    const int tl = blockIdx.y;
    const int64_t off = tree_off[t0 + tl];
    const int nl = (int)(tree_off[t0 + tl + 1] - off);
    const int m = nl - 1;
    const int nbk = (nl + 63) / 64;
    // blockIdx.x enumerates the pairs (I <= J) of 64-position blocks row by row
    int I = 0, rest = blockIdx.x;
    while (I < nbk && rest >= nbk - I) {
        rest -= nbk - I;
        ++I;
    }
    if (I >= nbk) return;
    const int J = I + rest;
    __shared__ int s_ta[64], s_tb[64];
    const int tid = threadIdx.x;
    if (tid < 64) s_ta[tid] = I * 64 + tid < nl ? leaf_taxon[off + I * 64 + tid] : -1;
    else if (tid < 128) s_tb[tid - 64] = J * 64 + tid - 64 < nl ? leaf_taxon[off + J * 64 + tid - 64] : -1;
    __syncthreads();
    const double *st = stv + st_off[tl];
    const int bj = tid & 63;
    const int pb = J * 64 + bj;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int ai = (tid >> 6) + 4 * q;
        const int pa = I * 64 + ai;
        if (pa >= pb || pb >= nl) continue;
        const double v = rmq_min<double>(st, m, pa, pb);  // 0 when the root separates the two
        if (v != 0.0) {
            const int ta = s_ta[ai], tb2 = s_tb[bj];
            atomicAdd(&w[(int64_t)ta * ld + tb2], v);
            atomicAdd(&w[(int64_t)tb2 * ld + ta], v);
        }
    }
}

// ---------------------------------------------------------------------------
// shared multi-rank build: gathered upper-triangle tiles -> this rank's rows of W
// ---------------------------------------------------------------------------
// Tile t of the global upper-triangle list was computed by rank t % world into slot
// t / world of that rank's chunk of `gathered`.  A rank copies the cells of every tile
// that fall into its rows, and -- for cells no tile owns directly -- the transposed
// image of the tile's cells whose COLUMN falls into its rows.
// (blockDim.x = tcw, the tile width of the build: 256, or 384 for the monotone kernel)
__global__ __launch_bounds__(1024) void k_unpack_tiles(const double *__restrict__ gathered,
                                                      const int2 *__restrict__ tiles, int world,
                                                      int64_t chunk_doubles, int n, int row_begin,
                                                      int row_end, double *__restrict__ w,
                                                      int64_t ld) {
    const int tcw = blockDim.x;
    const int t = blockIdx.x;
    const int2 tile = tiles[t];
    const double *src = gathered + (int64_t)(t % world) * chunk_doubles +
                        (int64_t)(t / world) * SCS_TR * tcw;
    const int c = tile.y * tcw + threadIdx.x;  // global column of this thread
    if (c >= n) return;
    const int r0 = tile.x * SCS_TR;
    const bool col_is_my_row = c >= row_begin && c < row_end;
    for (int i = 0; i < SCS_TR; ++i) {
        const int r = r0 + i;
        if (r >= n) break;
        const double v = src[i * tcw + threadIdx.x];
        if (r >= row_begin && r < row_end) w[(int64_t)(r - row_begin) * ld + c] = v;
        // cell (c, r) has no tile of its own iff its column group ends at or before its row block
        if (col_is_my_row && ((r / tcw) + 1) * tcw <= (c / SCS_TR) * SCS_TR)
            w[(int64_t)(c - row_begin) * ld + r] = v;
    }
}

// targeted exchange: copy the tiles a peer needs into its contiguous share of the send buffer
// (entry e: slot src_slot[e] of this rank's packed tiles -> position e of the send buffer)
__global__ __launch_bounds__(256) void k_pack_tiles(const double *__restrict__ tile_out,
                                                    const int32_t *__restrict__ src_slot, int tcw,
                                                    double *__restrict__ sendbuf) {
    const double2 *src = (const double2 *)(tile_out + (int64_t)src_slot[blockIdx.x] * SCS_TR * tcw);
    double2 *dst = (double2 *)(sendbuf + (int64_t)blockIdx.x * SCS_TR * tcw);
    for (int q = threadIdx.x; q < SCS_TR * tcw / 2; q += 256) dst[q] = src[q];
}

// received tiles (entry e of recvbuf is tile tiles[e]) -> this rank's rows of W: the cells of
// a tile that fall into its rows and the mirror image of the cells whose COLUMN does
__global__ __launch_bounds__(1024) void k_unpack_received(const double *__restrict__ recvbuf,
                                                         const int2 *__restrict__ tiles, int n,
                                                         int row_begin, int row_end,
                                                         double *__restrict__ w, int64_t ld) {
    const int tcw = blockDim.x;
    const int2 tile = tiles[blockIdx.x];
    const double *src = recvbuf + (int64_t)blockIdx.x * SCS_TR * tcw;
    const int c = tile.y * tcw + threadIdx.x;  // global column of this thread
    if (c >= n) return;
    const int r0 = tile.x * SCS_TR;
    const bool col_is_my_row = c >= row_begin && c < row_end;
    for (int i = 0; i < SCS_TR; ++i) {
        const int r = r0 + i;
        if (r >= n) break;
        const double v = src[i * tcw + threadIdx.x];
        if (r >= row_begin && r < row_end) w[(int64_t)(r - row_begin) * ld + c] = v;
        // cell (c, r) has no tile of its own iff its column group ends at or before its row block
        if (col_is_my_row && ((r / tcw) + 1) * tcw <= (c / SCS_TR) * SCS_TR)
            w[(int64_t)(c - row_begin) * ld + r] = v;
    }
}

// ---------------------------------------------------------------------------
// contraction: consecutive index ranges -> one vertex, weight = max over members
// ---------------------------------------------------------------------------
__global__ void k_contract(const double *__restrict__ w, int64_t ld, int old_row_begin,
                           const int32_t *__restrict__ gstart, int g_begin, int g_end, int n_groups,
                           double *__restrict__ out, int64_t ld_out) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;  // new column
    if (h >= n_groups) return;
    const int c0 = gstart[h], c1 = gstart[h + 1];
    // new rows (global), grid-stride: grid.y is capped at 65535
    for (int g = g_begin + blockIdx.y; g < g_end; g += gridDim.y) {
        double best = 0.0;
        if (h != g) {
            const int r0 = gstart[g], r1 = gstart[g + 1];
            best = w[(int64_t)(r0 - old_row_begin) * ld + c0];
            for (int r = r0; r < r1; ++r)
                for (int c = c0; c < c1; ++c) {
                    const double v = w[(int64_t)(r - old_row_begin) * ld + c];
                    best = v > best ? v : best;
                }
        }
        out[(int64_t)(g - g_begin) * ld_out + h] = best;
    }
}

// ---------------------------------------------------------------------------
// degrees: one wave per row
// ---------------------------------------------------------------------------
// IMG (round 5): the pass also leaves the single-precision image W32 of what the symmetric SYMM streams
// -- a row from the first column of its 512-column diagonal tile to the end of the padding -- for the
// eigen-solver's operator applications to its search directions (scs_eig.hip); same leading dimension.
template <bool IMG>
__global__ __launch_bounds__(256) void k_degrees(const double *__restrict__ w, int64_t ld, int n,
                                                  int rows, int row_begin,
                                                  double *__restrict__ deg_full,
                                                  float *__restrict__ w32 = nullptr, int w32_full = 0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const double *row = w + (int64_t)r * ld;
    double s0 = 0.0, s1 = 0.0;
    const int n2 = n & ~1;
    // (one device: the image is streamed in 128 x 512 tiles from the diagonal tile on; a row-partitioned rank
    // streams whole rows -- w32_full)
    const int c0 = w32_full ? 0 : (row_begin + r) / 512 * 512;
    float *row32 = IMG ? w32 + (int64_t)r * ld : nullptr;
    for (int j = lane * 2; j < n2; j += 128) {
        const double2 v = *(const double2 *)(row + j);
        s0 += v.x;
        s1 += v.y;
        if (IMG && j >= c0) *(float2 *)(row32 + j) = make_float2((float)v.x, (float)v.y);
    }
    if (IMG)  // the last column of an odd n and the padding (zeros)
        for (int j = n2 + lane; j < ld; j += 64) row32[j] = (float)row[j];
    if ((n & 1) && lane == 0) s0 += row[n - 1];
    double s = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) deg_full[row_begin + r] = s;
}

// SCS_BUILD_UPPER graphs (a rank stores the tiles on and right of the diagonal of its rows):
// the degree of v is the sum of what the symmetric SYMM applies -- row v from the first column
// of its 256-column diagonal tile on (the tile is stored whole: pairs inside it are met from
// both rows), plus, transposed, column v over the rows of earlier 256-row blocks.  One wave per
// row, then one thread per column walking this rank's rows in order: fixed orders, no atomics.
__global__ __launch_bounds__(256) void k_degrees_upper_rows(const double *__restrict__ w, int64_t ld, int n,
                                                             int rows, int row_begin, int col0,
                                                             double *__restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int rl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rl >= rows) return;
    const int r = row_begin + rl;
    const int c0 = r / 256 * 256;
    const double *row = w + (int64_t)rl * ld + (c0 - col0);
    const int cnt = n - c0;
    double s0 = 0.0, s1 = 0.0;
    const int n2 = cnt & ~1;
    for (int j = lane * 2; j < n2; j += 128) {
        const double2 v = *(const double2 *)(row + j);
        s0 += v.x;
        s1 += v.y;
    }
    if ((cnt & 1) && lane == 0) s0 += row[cnt - 1];
    double s = s0 + s1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) part[r] = s;
}

__global__ __launch_bounds__(256) void k_degrees_upper_cols(const double *__restrict__ w, int64_t ld, int n,
                                                             int row_begin, int row_end, int col0,
                                                             double *__restrict__ part) {
    const int c = col0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const int r_end = min(row_end, c / 256 * 256);  // rows of earlier 256-row blocks only
    double s = 0.0;
    for (int r = row_begin; r < r_end; ++r) s += w[(int64_t)(r - row_begin) * ld + (c - col0)];
    if (r_end > row_begin) part[c] += s;  // (c in this rank's rows: after its row sum; same thread order every run)
}

// row-partitioned graphs: the gathered per-rank degree vectors (zero outside a rank's rows)
__global__ void k_combine_degrees(const double *__restrict__ gathered, int world, int n,
                                  double *__restrict__ deg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = 0.0;
    for (int r = 0; r < world; ++r) d += gathered[(int64_t)r * n + i];
    deg[i] = d;
}

__global__ void k_dinv(const double *__restrict__ deg, int n, double *__restrict__ dinv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = deg[i];
    dinv[i] = d == 0.0 ? 1.0 : 1.0 / sqrt(d);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
namespace {

struct dev_buf {
    scs_ctx *ctx;
    void *p = nullptr;
    explicit dev_buf(scs_ctx *c) : ctx(c) {}
    ~dev_buf() {
        if (p) scs_dev_free(p);
    }
    int alloc(size_t bytes) {
        if (p) scs_dev_free(p);
        p = nullptr;
        SCS_HIP_CHECK(scs_dev_malloc(ctx, &p, bytes ? bytes : 16));
        return SCS_OK;
    }
};

// a device block from the context's cache (scs_block_alloc): the multi-GB exchange buffers of a
// shared build are reused from step to step instead of a hipMalloc / hipFree pair each
struct cached_buf {
    scs_ctx *ctx;
    void *p = nullptr;
    explicit cached_buf(scs_ctx *c) : ctx(c) {}
    ~cached_buf() {
        if (p) scs_block_release(ctx, p);
    }
    int alloc(size_t bytes) {
        if (p) scs_block_release(ctx, p);
        p = nullptr;
        return scs_block_alloc(ctx, bytes, &p);
    }
};

// a scratch buffer that lives in the context between calls (see scs_ctx::scratch)
struct pooled_buf {
    scs_ctx *ctx = nullptr;
    int slot = 0;
    void *p = nullptr;
    pooled_buf(scs_ctx *c, int s) : ctx(c), slot(s) {}
    ~pooled_buf() {
        auto &sl = ctx->scratch[slot];
        if (sl.cap > SCS_SCRATCH_KEEP) {
            scs_dev_free(sl.p);
            sl.p = nullptr;
            sl.cap = 0;
        }
    }
    int alloc(size_t bytes) {
        auto &sl = ctx->scratch[slot];
        if (bytes < 16) bytes = 16;
        if (sl.cap < bytes) {
            if (sl.p) scs_dev_free(sl.p);
            sl.p = nullptr;
            sl.cap = 0;
            SCS_HIP_CHECK(scs_dev_malloc(ctx, &sl.p, bytes));
            sl.cap = bytes;
        }
        p = sl.p;
        return SCS_OK;
    }
};

struct ev_pair {
    hipEvent_t a = nullptr, b = nullptr;
    ~ev_pair() {
        if (a) hipEventDestroy(a);
        if (b) hipEventDestroy(b);
    }
    int init() {
        SCS_HIP_CHECK(hipEventCreate(&a));
        SCS_HIP_CHECK(hipEventCreate(&b));
        return SCS_OK;
    }
};

int levels_for(int64_t m) {
    int l = 0;
    while (((int64_t)1 << l) <= m) ++l;
    return l;  // number of levels k with 2^k <= m
}

}  // namespace

static int graph_alloc(scs_ctx *ctx, int32_t n, int32_t row_begin, int32_t row_end,
                       hipStream_t stream, scs_graph **out, int32_t col0 = 0) {
    auto *g = new scs_graph();
    g->n = n;
    g->row_begin = row_begin;
    g->row_end = row_end;
    g->col0 = col0;
    g->upper = col0 > 0 || false;
    // k_symm's layout contract: ld a multiple of 512 doubles, padding columns zero
    g->ld = scs_round_up(n - col0, SCS_LD_ALIGN);
    const size_t rows = (size_t)(row_end - row_begin);
    size_t bytes = rows * (size_t)g->ld * sizeof(double);
    if (bytes < 16) bytes = 16;
    // (round 6) W of every size is a block of the device's arena: released, it serves whatever comes next -- a
    // level's forests, another context's W -- instead of waiting in this context for a graph of its own size
    {
        const int rc = scs_block_alloc(ctx, bytes, (void **)&g->d_w);
        if (rc != SCS_OK) {
            delete g;
            return rc;
        }
        g->w_bytes = bytes;
        g->w_block = true;
    }
    if (g->ld > n - col0) {
        hipError_t e = hipMemset2DAsync(g->d_w + (n - col0), (size_t)g->ld * 8, 0,
                                        (size_t)(g->ld - (n - col0)) * 8, rows, stream);
        if (e != hipSuccess) {
            if (g->w_block) scs_block_release(ctx, g->d_w);
            else scs_dev_free(g->d_w);
            delete g;
            scs_set_error("cannot clear the padding of W: %s", hipGetErrorString(e));
            return SCS_EHIP;
        }
    }
    *out = g;
    return SCS_OK;
}

extern "C" int scs_graph_free(scs_ctx *ctx, scs_graph *g) {
    if (!g) return SCS_OK;
    if (ctx) hipSetDevice(ctx->device);
    if (g->mf) {
        if (ctx) hipStreamSynchronize(ctx->stream);
        scs_matfree_release(g);
    }
    if (ctx && g->d_w && g->w_block) {
        // (stream order protects the block: whoever takes it next enqueues behind this graph's kernels -- or,
        // on another stream of the context, behind a synchronisation of its own, as every cached block)
        scs_block_release(ctx, g->d_w);
    } else if (ctx && g->d_w) {
        // keep the larger of the two buffers for the next graph; the kernels that used this
        // one are ordered before any later use by the context's stream
        if (!ctx->w_cache || g->w_bytes >= ctx->w_cache_bytes) {
            if (ctx->w_cache) scs_dev_free(ctx->w_cache);
            ctx->w_cache = g->d_w;
            ctx->w_cache_bytes = g->w_bytes;
        } else {
            scs_dev_free(g->d_w);
        }
    } else {
        scs_dev_free(g->d_w);
    }
    // (the two V-vectors come from the context's block cache: a hipFree each cost every step of the
    // benchmark and every node of a recursion a device-wide synchronisation)
    if (ctx && g->deg_stage) {
        // (an error between the two halves of scs_graph_prepare_degrees: the copy may still be running)
        hipStreamSynchronize(ctx->stream);
        scs_pinned_release(ctx, g->deg_stage);
    }
    if (ctx) {
        if (g->d_w32) {
            std::lock_guard<std::mutex> lock(ctx->cache_mu);
            if (scs_dbg("SCS_TRACE_SOLVES") && atoi(scs_dbg("SCS_TRACE_SOLVES")) && ctx->w32_cache)
                fprintf(stderr, "[image] hipFree of %.2f GB\n",
                        std::min(g->w32_bytes, ctx->w32_cache_bytes) / 1073741824.0);
            scs_dev_free(g->d_w32);  // (back to the arena: the next image of this context finds it there)
        }
        if (g->d_deg) scs_block_release(ctx, g->d_deg);
        if (g->d_dinv) scs_block_release(ctx, g->d_dinv);
    } else {
        scs_dev_free(g->d_w32);
        scs_dev_free(g->d_deg);
        scs_dev_free(g->d_dinv);
    }
    delete g;
    return SCS_OK;
}

extern "C" int scs_graph_shape(const scs_graph *g, int32_t *n, int32_t *rb, int32_t *re) {
    SCS_REQUIRE(g != nullptr, "scs_graph_shape: null graph");
    if (n) *n = g->n;
    if (rb) *rb = g->row_begin;
    if (re) *re = g->row_end;
    return SCS_OK;
}

int scs_gather_row_splits(scs_ctx *ctx, int32_t row_begin, int32_t row_end, int32_t n,
                          std::vector<int32_t> &splits) {
    const int world = ctx->comm.world;
    splits.assign(world + 1, 0);
    if (world == 1) {
        splits[0] = row_begin;
        splits[1] = row_end;
        return SCS_OK;
    }
    dev_buf send(ctx), recv(ctx);
    SCS_TRY(send.alloc(8));
    SCS_TRY(recv.alloc(8 * (size_t)world));
    double v = (double)row_begin;
    SCS_HIP_CHECK(hipMemcpyAsync(send.p, &v, 8, hipMemcpyHostToDevice, ctx->stream));
    SCS_TRY(scs_comm_allgather_f64(&ctx->comm, (const double *)send.p, (double *)recv.p, 1, ctx->stream));
    std::vector<double> h(world);
    SCS_HIP_CHECK(hipMemcpyAsync(h.data(), recv.p, 8 * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < world; ++r) splits[r] = (int32_t)h[r];
    splits[world] = n;
    for (int r = 0; r < world; ++r)
        SCS_REQUIRE(splits[r] < splits[r + 1], "row partition is not contiguous by rank");
    SCS_REQUIRE(splits[0] == 0 && splits[ctx->comm.rank] == row_begin && splits[ctx->comm.rank + 1] == row_end,
                "row partition does not tile [0, V)");
    return SCS_OK;
}

// ---- partial-coverage forests (round 5): which trees touch a tile at all
// The reference does one update per PAIR OF LEAVES of a tree (scs.py:644-658); the tile kernels walk
// every tree of a batch in every tile, adding +0.0 where a tree holds no row of the tile's row block or
// no column of its column group.  With trees that cover a few per cent of the taxa most (tile, tree)
// steps are such no-ops: every tile gets the list of the trees that do touch it, in tree order
// (k_accumulate_mono<.., .., true> walks the list; skipping an addend of +0.0 changes no bit).
// pcol[cg][t] = 1 when tree t0 + t holds a column of column group cg
__global__ __launch_bounds__(64) void k_col_presence(const int32_t *__restrict__ pos, int64_t npad, int n,
                                                      int cols_per_group, int n_batch,
                                                      unsigned char *__restrict__ pcol) {
    const int cg = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    bool any = false;
    for (int c = cg * cols_per_group + lane; c < min(n, (cg + 1) * cols_per_group); c += 64)
        any = any || pos[(int64_t)t * npad + c] >= 0;
    if (__ballot(any) != 0 && lane == 0) pcol[(int64_t)cg * n_batch + t] = 1;
    if (__ballot(any) == 0 && lane == 0) pcol[(int64_t)cg * n_batch + t] = 0;
}

// one thread per tile: the batch's trees with a present row in the tile's row block (the record's
// count) and a present column in its column group
__global__ void k_tile_lists(const int2 *__restrict__ tiles, int n_tiles, const unsigned char *__restrict__ rec,
                             int rec_bytes, int cnt_off, int n_batch, const unsigned char *__restrict__ pcol,
                             int32_t *__restrict__ lists, int32_t *__restrict__ list_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles) return;
    const int2 tile = tiles[i];
    const unsigned char *rb = rec + (int64_t)tile.x * n_batch * rec_bytes + cnt_off;
    const unsigned char *pc = pcol + (int64_t)tile.y * n_batch;
    int32_t *out = lists + (int64_t)i * n_batch;
    int c = 0;
    for (int t = 0; t < n_batch; ++t)
        if (pc[t] && *(const int *)(rb + (int64_t)t * rec_bytes) > 0) out[c++] = t;
    list_cnt[i] = c;
}

extern "C" int scs_pcg_build(scs_ctx *ctx, const scs_tables *tb, int32_t row_begin,
                             int32_t row_end, int32_t flags, scs_graph **out,
                             scs_build_stats *stats) {
    SCS_REQUIRE(ctx && tb && out, "scs_pcg_build: null argument");
    SCS_REQUIRE((flags & ~(SCS_BUILD_MONOTONE | SCS_BUILD_SHARED | SCS_BUILD_UPPER | SCS_BUILD_SCATTER)) == 0,
                "scs_pcg_build: unknown flag bits 0x%x", flags);
    const bool scatter = (flags & SCS_BUILD_SCATTER) != 0;
    // (the comparison variant indexes W by taxon ids straight from the tables: every chunk has to
    // have arrived and passed its range check first)
    if (scatter && tb) SCS_TRY(scs_tables_finish(ctx, tb));
    if (scatter) {
        if (!(flags & SCS_BUILD_MONOTONE) || (flags & (SCS_BUILD_SHARED | SCS_BUILD_UPPER)) ||
            row_begin != 0 || row_end != tb->n_taxa || ctx->comm.world != 1) {
            scs_set_error("scs_pcg_build: SCS_BUILD_SCATTER (the atomic comparison variant) takes monotone "
                          "tables, the whole matrix, one rank");
            return SCS_EUNSUP;
        }
    }
    SCS_REQUIRE((flags & (SCS_BUILD_SHARED | SCS_BUILD_UPPER)) != (SCS_BUILD_SHARED | SCS_BUILD_UPPER),
                "scs_pcg_build: SCS_BUILD_SHARED and SCS_BUILD_UPPER exclude each other");
    const bool monotone = (flags & SCS_BUILD_MONOTONE) != 0 &&
                          !(scs_dbg("SCS_NO_MONOTONE") && atoi(scs_dbg("SCS_NO_MONOTONE")));
    const int n = tb->n_taxa;
    SCS_REQUIRE(row_begin >= 0 && row_begin < row_end && row_end <= n,
                "scs_pcg_build: bad row range [%d, %d) for %d taxa", row_begin, row_end, n);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int world = ctx->comm.world, rank = ctx->comm.rank;
    // monotone weighting: tables over values (scs_mono.h); else (depth, value) pairs (scs_gen.h)
    const int cols_per_tile = MONO_TCW;
    static_assert(MONO_TCW == SCS_TCW, "one tile width for all kernels");
    const size_t entry_bytes = monotone ? 8 : sizeof(gap_entry);
    size_t rec_bytes = monotone ? R3_BYTES : G3_BYTES;
    const int n_cgroups = (n + cols_per_tile - 1) / cols_per_tile;
    // (the last column group may reach past n: its threads read "absent" positions)
    const int64_t npad = scs_round_up((int64_t)n_cgroups * cols_per_tile, SCS_NPAD);
    // Shared build (world > 1): the ranks split the upper-triangle tiles of the WHOLE matrix
    // round-robin, all-gather them and each unpacks its own rows -- every cell is computed
    // once across the job, as in the single-GPU symmetric schedule.  The decision depends
    // only on n and world, so every rank takes the same branch.
    bool shared = (flags & SCS_BUILD_SHARED) != 0 && world > 1;
    if (shared) {
        const int64_t nb_all = (n + SCS_TR - 1) / SCS_TR;
        const double tile_bytes = 0.5 * (double)nb_all * n_cgroups * SCS_TR * cols_per_tile * 8.0;
        if (tile_bytes * (1.0 + 1.0 / world) > 96.0 * 1024 * 1024 * 1024) shared = false;
    }
    // rows the accumulate kernels see: the whole matrix when shared
    const int b_row_begin = shared ? 0 : row_begin, b_row_end = shared ? n : row_end;
    const bool sym = !shared && (row_begin == 0 && row_end == n && world == 1);
    // SCS_BUILD_UPPER: only the tiles on and right of the diagonal of this rank's rows, stored
    // from column row_begin on; no mirror image, no exchange
    const bool trapezoid = (flags & SCS_BUILD_UPPER) != 0;
    if (trapezoid)
        SCS_REQUIRE(row_begin % cols_per_tile == 0,
                    "scs_pcg_build: SCS_BUILD_UPPER needs row_begin (%d) to be a multiple of %d", row_begin,
                    cols_per_tile);
    const bool upper = sym || shared || trapezoid;
    const int gb0 = b_row_begin / SCS_TR;  // global index of the first row block (upper test)
    const int rows = row_end - row_begin;
    const int n_blocks = (b_row_end - b_row_begin + SCS_TR - 1) / SCS_TR;

    scs_graph *g = nullptr;
    SCS_TRY(graph_alloc(ctx, n, row_begin, row_end, s, &g, trapezoid ? row_begin : 0));
    g->upper = trapezoid;
    if (scatter) SCS_HIP_CHECK(hipMemsetAsync(g->d_w, 0, (size_t)rows * (size_t)g->ld * 8, s));
    // (the kernels index W by global column: a base shifted left by col0 makes
    // w[(r - row_begin) * ld + c] land on the stored cell for every c >= col0)
    double *w_base = g->d_w - g->col0;
    struct guard {
        scs_ctx *c;
        scs_graph *g;
        ~guard() {
            if (g) scs_graph_free(c, g);
        }
    } gd{ctx, g};

    // ---- tile list
    std::vector<int2> tiles;
    tiles.reserve((size_t)n_blocks * n_cgroups);
    for (int b = 0; b < n_blocks; ++b)
        for (int c = 0; c < n_cgroups; ++c) {
            if (upper && (int64_t)(c + 1) * cols_per_tile <= (int64_t)(gb0 + b) * SCS_TR) continue;
            tiles.push_back(make_int2(b, c));
        }
    // XCD-aware order: workgroups go to the eight XCDs round-robin by index and all tiles
    // walk the trees at about the same pace.  Handing XCD x the row blocks b = x (mod 8), one
    // row block after the other, makes the ~96 workgroups that share an L2 read the same
    // block records and the same row-side table entries at the same time (measured: -4 % at
    // 10 000 leaves, -3 % at 50 000).  SCS_TILE_ORDER=0 keeps the plain row-major order.
    // In a shared multi-rank build the job-wide list is dealt round-robin to the ranks first
    // and every rank's share is ordered for its own XCDs (all ranks compute all shares, so
    // entry world * k + r of the list is still rank r's k-th tile).
    {
        const bool per_xcd = !(scs_dbg("SCS_TILE_ORDER") && atoi(scs_dbg("SCS_TILE_ORDER")) == 0);
        auto xcd_reorder = [](std::vector<int2> &v) {
            if (v.size() <= 8) return;
            std::vector<int2> per[8];
            for (const int2 &t : v) per[t.x & 7].push_back(t);  // stays (row, column) sorted
            std::vector<int2> ordered;
            ordered.reserve(v.size());
            size_t at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            while (ordered.size() < v.size())
                for (int x = 0; x < 8; ++x)
                    if (at[x] < per[x].size()) ordered.push_back(per[x][at[x]++]);
            v.swap(ordered);
        };
        if (per_xcd && !shared) {
            xcd_reorder(tiles);
        } else if (per_xcd) {
            std::vector<std::vector<int2>> share(world);
            for (size_t i = 0; i < tiles.size(); ++i) share[i % world].push_back(tiles[i]);
            for (auto &v : share) xcd_reorder(v);
            for (int r = 0; r < world; ++r)
                for (size_t k = 0; k < share[r].size(); ++k) tiles[(size_t)world * k + r] = share[r][k];
        }
    }
    // shared: tile i of the job-wide list belongs to rank i % world and lands in slot
    // i / world of that rank's packed buffer
    std::vector<int2> all_tiles;
    dev_buf d_all_tiles(ctx);
    cached_buf d_tile_out(ctx), d_gathered(ctx);
    size_t chunk_doubles = 0;
    // the exchange of a shared build: point to point by default, one all-gather on request
    const bool exchange_allgather =
        scs_dbg("SCS_EXCHANGE") && std::string(scs_dbg("SCS_EXCHANGE")) == "allgather";
    if (shared) {
        all_tiles.swap(tiles);
        for (size_t i = rank; i < all_tiles.size(); i += world) tiles.push_back(all_tiles[i]);
        const size_t slots = (all_tiles.size() + world - 1) / world;
        chunk_doubles = slots * SCS_TR * cols_per_tile;
        SCS_TRY(d_all_tiles.alloc(all_tiles.size() * sizeof(int2)));
        SCS_HIP_CHECK(hipMemcpyAsync(d_all_tiles.p, all_tiles.data(),
                                     all_tiles.size() * sizeof(int2), hipMemcpyHostToDevice, s));
        SCS_TRY(d_tile_out.alloc(chunk_doubles * 8));
        if (exchange_allgather) SCS_TRY(d_gathered.alloc(chunk_doubles * 8 * world));
        if (tiles.size() < slots)  // the unused last slot is gathered too: keep it defined
            SCS_HIP_CHECK(hipMemsetAsync((double *)d_tile_out.p + (slots - 1) * SCS_TR * cols_per_tile,
                                         0, (size_t)SCS_TR * cols_per_tile * 8, s));
    }
    pooled_buf d_tiles(ctx, 0);
    SCS_TRY(d_tiles.alloc(std::max<size_t>(tiles.size(), 1) * sizeof(int2)));
    SCS_HIP_CHECK(hipMemcpyAsync(d_tiles.p, tiles.data(), tiles.size() * sizeof(int2),
                                 hipMemcpyHostToDevice, s));

    // ---- tree-parallel build (scs_mono.h, k_sum_tree_tiles): a node of a few tiles and many trees --
    // the mid-size nodes of the recursion -- gives every (tile, tree) pair its own workgroup and adds
    // the trees' cells up in order afterwards.  SCS_TREE_PARALLEL=0 / 1 force either way (tests).
    // Measured (tools/node_profile2.py, 5 000 trees): 200 taxa 10.8 -> 1.2 ms, 400 taxa (11 tiles) 11 ->
    // 3.1 ms = 0.056 us per tile and tree against 1.9 us per tree for the producer / consumer walk:
    // up to 24 tiles.
    bool tree_par = sym && !scatter && !tiles.empty() && tiles.size() <= 24 && tb->n_trees >= 128;
    if (const char *e = scs_dbg("SCS_TREE_PARALLEL")) tree_par = sym && !scatter && !tiles.empty() && atoi(e) != 0;
    const size_t cells_per_tree = tiles.size() * (size_t)SCS_TR * cols_per_tile * 8;

    // ---- producer / consumer workgroups (scs_mono_wide.h): two tiles of ONE row block per workgroup,
    // the row block's table expanded once for both, four more waves running the column step ahead
    // of the cells.  Worth it once the groups fill the chip (one twelve-wave workgroup per CU) --
    // measured with the long tree batches it prefers (below): -11 % at 10 000 leaves per tree, -12 %
    // at 50 000; a handful of tiles -- a node of the deep recursion -- keeps one workgroup per tile.
    // SCS_WIDE=0 / 1 force either kernel (A/B runs, tests).
    constexpr int PIPE_NG = 2;
    // (Round 4, later: it also pays on a few dozen tiles -- a node of 1 000 to 5 000 taxa in the
    // recursion, where the walk is bound by the latency of a tree's step, not by the chip:
    // 1 000 taxa x 5 000 trees 11.1 -> 9.5 ms, 3 000 x 5 000 15.5 -> 11.2, 3 000 x 300 1.03 -> 0.78;
    // up to 24 tiles the tree-parallel build below is faster still when the trees are many.)
    // (diagnostic switches are read on every call: tests and A/B tools set them in-process)
    const size_t wide_min_tiles = 13;
    // Partial-coverage forests: when an average tree holds less than 1 / 64 of a tile's 64 + 256 rows
    // and columns' worth of the taxa -- less than about 1.5 % of them -- most (tile, tree) steps add
    // +0.0 everywhere; every tile then walks its own list of trees (k_tile_lists; the 4-wave kernel:
    // its step is the cheapest to skip).  SCS_TILE_LISTS=0 / 1 force either way.
    bool listed = monotone && !scatter && !tiles.empty() && tb->n_trees > 0 &&
                  (double)tb->n_leaves / ((double)tb->n_trees * std::max(n, 1)) < 1.0 / 64.0;
    if (const char *e = scs_dbg("SCS_TILE_LISTS")) listed = monotone && !scatter && !tiles.empty() && atoi(e) != 0;
    if (listed) tree_par = false;
    int wide_mode = (monotone && !scatter && !tree_par && !listed && tiles.size() >= wide_min_tiles) ? 3 : 0;
    bool wide_forced = false;  // an explicit SCS_WIDE=1 also lifts the trees-per-batch gate below
    if (const char *e = scs_dbg("SCS_WIDE")) {
        wide_mode = (monotone && !scatter && !tiles.empty() && atoi(e)) ? 3 : 0;
        wide_forced = wide_mode != 0;
        if (wide_mode) listed = false;
    }
    // the producer / consumer kernel needs more dynamic LDS than every device offers
    if (wide_mode && ctx->max_lds_bytes < (int)spec_layout::LDS_BYTES) {
        wide_mode = 0;
        wide_forced = false;
    }
    if (wide_mode) tree_par = false;  // (SCS_WIDE=1 wins over a tree-parallel build)
    const bool wide = wide_mode != 0;
    const int group_tiles = PIPE_NG;
    std::vector<int4> groups;
    dev_buf d_groups(ctx);
    if (wide) {
        rec_bytes = wide_layout<PIPE_NG>::BYTES;
        // the tiles of a row block in list order, two at a time; XCD x is handed the groups of
        // the row blocks b = x (mod 8), one row block after the other (as the tile order above)
        std::vector<std::vector<int>> of_block((size_t)n_blocks);
        for (size_t i = 0; i < tiles.size(); ++i) of_block[(size_t)tiles[i].x].push_back((int)i);
        std::vector<int4> per[8];
        for (int b = 0; b < n_blocks; ++b) {
            const std::vector<int> &v = of_block[(size_t)b];
            for (size_t k = 0; k < v.size(); k += group_tiles)
                per[b & 7].push_back(make_int4(v[k], k + 1 < v.size() ? v[k + 1] : -1,
                                               (group_tiles > 2 && k + 2 < v.size()) ? v[k + 2] : -1, 0));
        }
        size_t total = 0, at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (auto &v : per) total += v.size();
        groups.reserve(total);
        while (groups.size() < total)
            for (int x = 0; x < 8; ++x)
                if (at[x] < per[x].size()) groups.push_back(per[x][at[x]++]);
        SCS_TRY(d_groups.alloc(groups.size() * sizeof(int4)));
        SCS_HIP_CHECK(hipMemcpyAsync(d_groups.p, groups.data(), groups.size() * sizeof(int4),
                                     hipMemcpyHostToDevice, s));
    }

    auto table_entries = [](int64_t m) -> int64_t { return (int64_t)levels_for(m) * m; };

    // ---- batch plan: bound range-minimum tables + records + positions by the workspace
    const int M = tb->n_trees;
    std::vector<int> batch_start{0};
    {
        // A batch is also capped in trees.  All tiles of a launch start at the batch's first tree
        // and drift apart as they walk the trees; the gathers into a tree's range-minimum table
        // hit the L2 (or at least the Infinity Cache) only while the workgroups are within a few
        // trees of each other.  A new launch re-synchronises them for the price of re-reading
        // and re-writing the tile sums (the mirror image is only written by the last batch), so
        // the bigger a tree's table, the shorter the batch: about 600 MB of tables per batch,
        // between 64 and 256 trees (measured: 10 000 leaves 256 trees -- 128 within 1 %;
        // 50 000 leaves: 96 / 80 trees 682 ms, 256 trees 720 ms, 32 trees 761 ms; 100 000
        // leaves: 48-96 trees 7.5 s, 256 trees 8.4 s).  SCS_BATCH_TREES overrides.
        // (Preparing batch k + 1 on a second stream while batch k accumulates was tried and is
        // slower -- 50 000 leaves: 785 ms instead of 697 ms at any stream priority: the
        // preparation kernels take CU slots a few at a time, the tiles fall out of step and the
        // new tables push the current ones out of the caches.  The batches stay sequential.)
        const int batch_trees_env = scs_dbg("SCS_BATCH_TREES") ? atoi(scs_dbg("SCS_BATCH_TREES")) : 0;
        int max_batch_trees = 256;
        {
            const double avg_leaves = (double)tb->n_leaves / std::max(M, 1);
            const double table_bytes =
                (double)table_entries((int64_t)std::max(avg_leaves - 1.0, 1.0)) * (double)entry_bytes;
            max_batch_trees = (int)std::min(256.0, std::max(64.0, 600e6 / std::max(table_bytes, 1.0)));
        }
        // (a node of the deep recursion -- a few hundred taxa, thousands of tiny trees -- is a
        // handful of tiles whose tables sit in L2: one batch, one set of launches)
        if (n <= 2048) max_batch_trees = 4096;
        // tables whose later trees are still on their way (page-locked source, scs_tables_upload): the
        // trees that have arrived make a first, short batch -- the copy of the rest runs beside it
        const int arrived = tb->late_start.empty() ? M : tb->late_start[0];
        if (wide) {
            // The producer / consumer kernel wants LONG batches: its range-minimum queries are off the
            // cells' critical path (the locality of a short batch buys it little), while every launch
            // costs it a prologue, tile loads and stores and a thin last round that no second
            // workgroup on the CU hides.  Measured (profiles/r04_spec_batch_sweep.txt), 50 000 leaves:
            // 93 trees per batch 705 ms, 186: 633, 256: 613, 400: 602, 667: 617, 1 000: 603, 2 000: 645
            // (the 4-wave kernel: 682 at 93, 715 at 186); 10 000 leaves: 250: 6.48 ms, 500: 6.2.
            // Up to 512 trees, the batches behind the first of equal length.
            const int rest = M - (arrived < M ? arrived : 0);
            const int k = std::max(1, (rest + 511) / 512);
            max_batch_trees = std::max(96, (rest + k - 1) / k);
        }
        // (tree-parallel: the cells of a batch's trees stay within the 1 GiB scratch the context keeps)
        if (tree_par)
            max_batch_trees = (int)std::max<size_t>(16, std::min<size_t>((size_t)max_batch_trees,
                                                                         SCS_SCRATCH_KEEP / cells_per_tree));
        if (batch_trees_env > 0) max_batch_trees = batch_trees_env;
        // (a forced tree-parallel build or batch length still keeps a batch's cells within the scratch)
        if (tree_par)
            max_batch_trees = (int)std::max<size_t>(1, std::min<size_t>((size_t)max_batch_trees,
                                                                        SCS_SCRATCH_KEEP / std::max<size_t>(cells_per_tree, 1)));
        size_t used = 0;
        for (int t = 0; t < M; ++t) {
            const int64_t nt = tb->h_tree_off[t + 1] - tb->h_tree_off[t];
            const int64_t m = nt - 1;
            const size_t need =
                (size_t)table_entries(m) * entry_bytes + (size_t)npad * 4 + (size_t)n_blocks * rec_bytes;
            if ((used + need > ctx->ws_limit || t - batch_start.back() >= max_batch_trees ||
                 (t == arrived && batch_start.size() == 1)) &&
                t > batch_start.back()) {
                batch_start.push_back(t);
                used = 0;
            }
            used += need;
        }
        batch_start.push_back(M);
    }
    const int n_batches = (int)batch_start.size() - 1;

    // timing events live in the context (created once)
    struct ev_ref {
        hipEvent_t a, b;
    } ev_total, ev_prep, ev_acc;
    for (auto &e : ctx->build_events)
        if (!e) SCS_HIP_CHECK(hipEventCreate(&e));
    ev_total = {ctx->build_events[0], ctx->build_events[1]};
    ev_prep = {ctx->build_events[2], ctx->build_events[3]};
    ev_acc = {ctx->build_events[4], ctx->build_events[5]};
    float prep_ms = 0.f, acc_ms = 0.f;
    SCS_HIP_CHECK(hipEventRecord(ev_total.a, s));

    int spec_batches = 0, spec_trees = 0;
    float spec_ms = 0.f;
    bool single_batch_spec = false;
    pooled_buf d_pos(ctx, 1), d_st(ctx, 3), d_stoff(ctx, 4), d_rec(ctx, 5), d_cells(ctx, 6);
    dev_buf d_pcol(ctx), d_lists(ctx), d_list_cnt(ctx);  // (partial-coverage forests: per-tile tree lists)
    int listed_batches = 0;
    for (int bi = 0; bi < n_batches; ++bi) {
        const int t0 = batch_start[bi], t1 = batch_start[bi + 1];
        const int nb = t1 - t0;
        std::vector<int64_t> st_off(nb + 1);
        int max_levels = 0;
        int64_t max_n = 0;
        st_off[0] = 0;
        for (int t = t0; t < t1; ++t) {
            const int64_t nt = tb->h_tree_off[t + 1] - tb->h_tree_off[t];
            st_off[t - t0 + 1] = st_off[t - t0] + table_entries(nt - 1);
            max_levels = std::max(max_levels, levels_for(nt - 1));
            max_n = std::max(max_n, nt);
        }
        const size_t need_pos = (size_t)nb * npad * 4;
        // +64: a tree without gaps may be probed at entry 0
        const size_t need_st = (size_t)st_off[nb] * entry_bytes + 64;
        const size_t need_stoff = (size_t)(nb + 1) * 8;
        const size_t need_rec = (size_t)n_blocks * nb * rec_bytes;
        // A short batch (the first 64 trees of tables still on their way, a forest's last few trees)
        // is the 4-wave kernel's: a twelve-wave workgroup has a CU to itself, nothing hides its
        // prologue, its tile stores and the thin last round of its launch, and below ~100 trees
        // that costs more than the shorter steps save (measured at 10 000 leaves: 64 + 218 + 218
        // trees 6.93 ms either way, 218 + 218 with the 4-wave kernel in front 6.5).
        const int wide_min_trees = wide_forced ? 1 : 96;
        // (with many rounds of workgroups per launch -- 50 000 leaves: 150 -- the prologue and the thin
        // last round are noise and the short first batch is the producer / consumer kernel's too:
        // 64 trees 25 -> 18 ms there)
        const bool wide_b = wide && (nb >= wide_min_trees || (groups.size() >= 8 * 256 && nb >= 16));
        if (wide_b) {
            ++spec_batches;
            spec_trees += nb;
            single_batch_spec = true;
        }
        SCS_TRY(d_pos.alloc(need_pos));
        SCS_TRY(d_st.alloc(need_st));
        SCS_TRY(d_stoff.alloc(need_stoff));
        SCS_TRY(d_rec.alloc(need_rec));

        SCS_HIP_CHECK(hipEventRecord(ev_prep.a, s));
        SCS_TRY(scs_tables_wait(ctx, tb, t1, s));  // (leaf arrays still on their way: scs_tables_upload)
        SCS_HIP_CHECK(hipMemcpyAsync(d_stoff.p, st_off.data(), need_stoff, hipMemcpyHostToDevice, s));
        SCS_HIP_CHECK(hipMemsetAsync(d_pos.p, 0xFF, need_pos, s));
        dim3 grid_l((unsigned)((max_n + 255) / 256), (unsigned)nb);
        if (monotone) {
            k_positions_values<<<grid_l, 256, 0, s>>>(tb->d_tree_off, tb->d_leaf_taxon,
                                                      tb->d_adj_depth, tb->d_adj_val, tb->d_tree_w,
                                                      t0, (int32_t *)d_pos.p, npad,
                                                      (const int64_t *)d_stoff.p, (double *)d_st.p);
            if (max_n <= SPARSE_FUSED_MAX_LEAVES && max_levels > 1)
                k_sparse_levels_fused<double><<<(unsigned)nb, 1024, 0, s>>>(
                    tb->d_tree_off, t0, (const int64_t *)d_stoff.p, (double *)d_st.p, pick_smaller_value{});
            else
            for (int k = 1; k < max_levels; ++k)
                k_sparse_level<double><<<grid_l, 256, 0, s>>>(tb->d_tree_off, t0, k,
                                                              (const int64_t *)d_stoff.p,
                                                              (double *)d_st.p);
            if (wide_b)
                k_block_records_wide<PIPE_NG><<<dim3((unsigned)n_blocks, (unsigned)nb), 64, 0, s>>>(
                    tb->d_tree_off, t0, nb, (const int32_t *)d_pos.p, npad, (const int64_t *)d_stoff.p,
                    (const double *)d_st.p, b_row_begin, b_row_end, (unsigned char *)d_rec.p);
            else if (!scatter)
            k_block_records_mono<<<dim3((unsigned)n_blocks, (unsigned)nb), 64, 0, s>>>(
                tb->d_tree_off, t0, nb, (const int32_t *)d_pos.p, npad, (const int64_t *)d_stoff.p,
                (const double *)d_st.p, b_row_begin, b_row_end, (unsigned char *)d_rec.p);
        } else {
            k_positions_pairs<<<grid_l, 256, 0, s>>>(tb->d_tree_off, tb->d_leaf_taxon,
                                                     tb->d_adj_depth, tb->d_adj_val, tb->d_tree_w,
                                                     t0, (int32_t *)d_pos.p, npad,
                                                     (const int64_t *)d_stoff.p, (gap_entry *)d_st.p);
            if (max_n <= SPARSE_FUSED_MAX_LEAVES && max_levels > 1)
                k_sparse_levels_fused<gap_entry><<<(unsigned)nb, 1024, 0, s>>>(
                    tb->d_tree_off, t0, (const int64_t *)d_stoff.p, (gap_entry *)d_st.p, pick_shallower_pair{});
            else
            for (int k = 1; k < max_levels; ++k)
                k_sparse_level_pairs<<<grid_l, 256, 0, s>>>(tb->d_tree_off, t0, k,
                                                            (const int64_t *)d_stoff.p,
                                                            (gap_entry *)d_st.p);
            k_block_records_gen<<<dim3((unsigned)n_blocks, (unsigned)nb), 64, 0, s>>>(
                tb->d_tree_off, t0, nb, (const int32_t *)d_pos.p, npad, (const int64_t *)d_stoff.p,
                (const gap_entry *)d_st.p, b_row_begin, b_row_end, (unsigned char *)d_rec.p);
        }
        SCS_HIP_CHECK(hipEventRecord(ev_prep.b, s));

        SCS_HIP_CHECK(hipEventRecord(ev_acc.a, s));
        const unsigned nt = (unsigned)tiles.size();
        if (scatter) {
            const int64_t nbk = (max_n + 63) / 64;
            k_scatter_atomic<<<dim3((unsigned)(nbk * (nbk + 1) / 2), (unsigned)nb), 256, 0, s>>>(
                tb->d_tree_off, tb->d_leaf_taxon, t0, (const int64_t *)d_stoff.p, (const double *)d_st.p,
                g->d_w, g->ld);
        } else if (monotone) {
            mono_params mp;
            mp.tiles = (const int2 *)d_tiles.p;
            mp.rec = (const unsigned char *)d_rec.p;
            mp.pos = (const int32_t *)d_pos.p;
            mp.npad = npad;
            mp.stv = (const double *)d_st.p;
            mp.n_batch = nb;
            mp.w = w_base;
            mp.ld = g->ld;
            mp.n = n;
            mp.row_begin = b_row_begin;
            mp.row_end = b_row_end;
            mp.load_w = bi > 0;
            mp.mirror = (sym && bi == n_batches - 1) ? 1 : 0;
            mp.tile_out = shared ? (double *)d_tile_out.p : nullptr;
            mp.stamps = nullptr;
            mp.split_tiles = 0;
            mp.lists = nullptr;
            mp.list_cnt = nullptr;
            const bool stamp = scs_dbg("SCS_ACC_STAMP") && atoi(scs_dbg("SCS_ACC_STAMP"));
            if (wide_b) {
                wide_params wp;
                wp.m = mp;
                wp.groups = (const int4 *)d_groups.p;
                wp.producer_prio = 2;
                const unsigned ng = (unsigned)groups.size();
                dev_buf d_st8(ctx);
                if (stamp) {
                    SCS_TRY(d_st8.alloc(64));
                    SCS_HIP_CHECK(hipMemsetAsync(d_st8.p, 0, 64, s));
                    wp.m.stamps = (unsigned long long *)d_st8.p;
                }
                // (more dynamic LDS than the default 64 KiB: per function and device, set every time)
#define SCS_LAUNCH_SPEC(SYM_, STAMP_)                                                                   \
    do {                                                                                                \
        SCS_HIP_CHECK(hipFuncSetAttribute((const void *)k_accumulate_spec<SYM_, STAMP_>,                \
                                          hipFuncAttributeMaxDynamicSharedMemorySize,                   \
                                          (int)spec_layout::LDS_BYTES));                                \
        k_accumulate_spec<SYM_, STAMP_><<<ng, spec_layout::THREADS, spec_layout::LDS_BYTES, s>>>(wp);   \
    } while (0)
                if (ng) {
                    if (stamp && sym) SCS_LAUNCH_SPEC(true, true);
                    else if (stamp) SCS_LAUNCH_SPEC(false, true);
                    else if (sym) SCS_LAUNCH_SPEC(true, false);
                    else SCS_LAUNCH_SPEC(false, false);
                }
#undef SCS_LAUNCH_SPEC
                if (stamp) {
                    unsigned long long h[8];
                    SCS_HIP_CHECK(hipMemcpyAsync(h, d_st8.p, 64, hipMemcpyDeviceToHost, s));
                    SCS_HIP_CHECK(hipStreamSynchronize(s));
                    // (every wave adds its own phases: a producer phase is averaged over all twelve waves
                    // -- times 3 for the producers' own figure, a consumer phase times 1.5)
                    const char *nm[5] = {"producer: pairs of the next tree", "producer: records + search + wait",
                                         "consumer: pair + table state", "consumer: cells + expansion", "barrier"};
                    double tot = 0;
                    for (int i = 0; i < 5; ++i) tot += (double)h[i];
                    for (int i = 0; i < 5; ++i)
                        fprintf(stderr, "[stamp spec] %-36s %6.2f %%  (%.0f cycles per wave-step)\n", nm[i],
                                100.0 * h[i] / tot, (double)h[i] / ((double)h[7] * nb));
                }
            } else if (nt && stamp) {
                dev_buf d_st8(ctx);
                SCS_TRY(d_st8.alloc(64));
                SCS_HIP_CHECK(hipMemsetAsync(d_st8.p, 0, 64, s));
                mp.stamps = (unsigned long long *)d_st8.p;
                if (sym) k_accumulate_mono<true, true><<<nt, MONO_TCW, 0, s>>>(mp);
                else k_accumulate_mono<false, true><<<nt, MONO_TCW, 0, s>>>(mp);
                unsigned long long h[8];
                SCS_HIP_CHECK(hipMemcpyAsync(h, d_st8.p, 64, hipMemcpyDeviceToHost, s));
                SCS_HIP_CHECK(hipStreamSynchronize(s));
                const char *nm[7] = {"wait loads+record", "barrier A", "combine", "search + issue",
                                     "expand", "barrier B", "cells"};
                double tot = 0;
                for (int i = 0; i < 7; ++i) tot += (double)h[i];
                for (int i = 0; i < 7; ++i)
                    fprintf(stderr, "[stamp] %-24s %6.2f %%  (%.0f cycles per wave-step)\n", nm[i],
                            100.0 * h[i] / tot, (double)h[i] / ((double)h[7] * nb));
            } else if (nt && tree_par) {
                SCS_TRY(d_cells.alloc((size_t)nb * cells_per_tree));
                mono_params ap = mp;  // the addends: one workgroup per (tile, tree), from zero, no W
                ap.split_tiles = (int)nt;
                ap.load_w = 0;
                ap.mirror = 0;
                ap.tile_out = (double *)d_cells.p;
                k_accumulate_mono<true, false><<<nt * (unsigned)nb, MONO_TCW, 0, s>>>(ap);
                k_sum_tree_tiles<true, mono_params><<<nt * SCS_TR, MONO_TCW, 0, s>>>(mp, (const double *)d_cells.p,
                                                                                    (int)nt);
            } else if (nt && listed) {
                // every tile's own list of the batch's trees (those that touch it), then the walk over it
                SCS_TRY(d_pcol.alloc((size_t)n_cgroups * nb));
                SCS_TRY(d_lists.alloc((size_t)nt * nb * 4));
                SCS_TRY(d_list_cnt.alloc((size_t)nt * 4));
                k_col_presence<<<dim3((unsigned)n_cgroups, (unsigned)nb), 64, 0, s>>>(
                    (const int32_t *)d_pos.p, npad, n, cols_per_tile, nb, (unsigned char *)d_pcol.p);
                k_tile_lists<<<(nt + 255) / 256, 256, 0, s>>>((const int2 *)d_tiles.p, (int)nt,
                                                             (const unsigned char *)d_rec.p, R3_BYTES, R3_CNT, nb,
                                                             (const unsigned char *)d_pcol.p, (int32_t *)d_lists.p,
                                                             (int32_t *)d_list_cnt.p);
                mp.lists = (const int32_t *)d_lists.p;
                mp.list_cnt = (const int32_t *)d_list_cnt.p;
                if (sym) k_accumulate_mono<true, false, true><<<nt, MONO_TCW, 0, s>>>(mp);
                else k_accumulate_mono<false, false, true><<<nt, MONO_TCW, 0, s>>>(mp);
                ++listed_batches;
            } else if (nt) {
                if (sym) k_accumulate_mono<true, false><<<nt, MONO_TCW, 0, s>>>(mp);
                else k_accumulate_mono<false, false><<<nt, MONO_TCW, 0, s>>>(mp);
            }
        } else {
            gen_params gp;
            gp.tiles = (const int2 *)d_tiles.p;
            gp.rec = (const unsigned char *)d_rec.p;
            gp.pos = (const int32_t *)d_pos.p;
            gp.npad = npad;
            gp.ste = (const gap_entry *)d_st.p;
            gp.n_batch = nb;
            gp.w = w_base;
            gp.ld = g->ld;
            gp.n = n;
            gp.row_begin = b_row_begin;
            gp.row_end = b_row_end;
            gp.load_w = bi > 0;
            gp.mirror = (sym && bi == n_batches - 1) ? 1 : 0;
            gp.tile_out = shared ? (double *)d_tile_out.p : nullptr;
            gp.split_tiles = 0;
            if (nt && tree_par) {
                SCS_TRY(d_cells.alloc((size_t)nb * cells_per_tree));
                gen_params ap = gp;  // the addends: one workgroup per (tile, tree), from zero, no W
                ap.split_tiles = (int)nt;
                ap.load_w = 0;
                ap.mirror = 0;
                ap.tile_out = (double *)d_cells.p;
                k_accumulate_gen<true><<<nt * (unsigned)nb, MONO_TCW, 0, s>>>(ap);
                k_sum_tree_tiles<true, gen_params><<<nt * SCS_TR, MONO_TCW, 0, s>>>(gp, (const double *)d_cells.p,
                                                                                   (int)nt);
            } else if (nt) {
                if (sym) k_accumulate_gen<true><<<nt, MONO_TCW, 0, s>>>(gp);
                else k_accumulate_gen<false><<<nt, MONO_TCW, 0, s>>>(gp);
            }
        }
        SCS_HIP_CHECK(hipGetLastError());
        SCS_HIP_CHECK(hipEventRecord(ev_acc.b, s));
        if (n_batches > 1) {
            // the one event pair is re-recorded by the next batch: read it now (a single
            // batch is read after the final synchronisation instead -- small builds are
            // dominated by host round trips)
            SCS_HIP_CHECK(hipEventSynchronize(ev_acc.b));
            float ms = 0.f;
            SCS_HIP_CHECK(hipEventElapsedTime(&ms, ev_prep.a, ev_prep.b));
            prep_ms += ms;
            SCS_HIP_CHECK(hipEventElapsedTime(&ms, ev_acc.a, ev_acc.b));
            acc_ms += ms;
            if (wide_b) spec_ms += ms;
        }
    }
    float exch_ms = 0.f;
    double exch_bytes = 0.0;
    if (shared && exchange_allgather) {
        ev_pair ev_x;
        SCS_TRY(ev_x.init());
        SCS_HIP_CHECK(hipEventRecord(ev_x.a, s));
        SCS_TRY(scs_comm_allgather_f64(&ctx->comm, (const double *)d_tile_out.p,
                                       (double *)d_gathered.p, chunk_doubles, s));
        k_unpack_tiles<<<(unsigned)all_tiles.size(), cols_per_tile, 0, s>>>(
            (const double *)d_gathered.p, (const int2 *)d_all_tiles.p, world,
            (int64_t)chunk_doubles, n, row_begin, row_end, g->d_w, g->ld);
        SCS_HIP_CHECK(hipGetLastError());
        SCS_HIP_CHECK(hipEventRecord(ev_x.b, s));
        SCS_HIP_CHECK(hipEventSynchronize(ev_x.b));
        SCS_HIP_CHECK(hipEventElapsedTime(&exch_ms, ev_x.a, ev_x.b));
        exch_bytes = (double)chunk_doubles * 8.0 * (world - 1);
    } else if (shared) {
        // ---- point-to-point exchange: a tile travels only to the ranks whose rows it touches
        // (directly: its row block; mirrored: its columns).  Every rank derives the same plan
        // from the job-wide tile list and the row splits.
        std::vector<int32_t> splits;
        SCS_TRY(scs_gather_row_splits(ctx, row_begin, row_end, n, splits));
        auto needs = [&](int dst, const int2 &t) {
            const int64_t lo = splits[dst], hi = splits[dst + 1];
            const int64_t r_lo = (int64_t)t.x * SCS_TR, r_hi = std::min<int64_t>(r_lo + SCS_TR, n);
            const int64_t c_lo = (int64_t)t.y * cols_per_tile, c_hi = std::min<int64_t>(c_lo + cols_per_tile, n);
            return (r_lo < hi && lo < r_hi) || (c_lo < hi && lo < c_hi);
        };
        std::vector<int64_t> send_off(world + 1, 0), recv_off(world + 1, 0);
        std::vector<int32_t> send_slots;  // per destination, in slot order
        std::vector<int2> recv_tiles;     // per source, in the source's slot order
        for (int p = 0; p < world; ++p) {
            for (size_t i = rank; i < all_tiles.size(); i += world)
                if (needs(p, all_tiles[i])) send_slots.push_back((int32_t)(i / world));
            send_off[p + 1] = (int64_t)send_slots.size() * SCS_TR * cols_per_tile;
            for (size_t i = p; i < all_tiles.size(); i += world)
                if (needs(rank, all_tiles[i])) recv_tiles.push_back(all_tiles[i]);
            recv_off[p + 1] = (int64_t)recv_tiles.size() * SCS_TR * cols_per_tile;
        }
        cached_buf d_send(ctx), d_recv(ctx);
        dev_buf d_slots(ctx), d_rtiles(ctx);
        SCS_TRY(d_send.alloc((size_t)send_off[world] * 8));
        SCS_TRY(d_recv.alloc((size_t)recv_off[world] * 8));
        SCS_TRY(d_slots.alloc(send_slots.size() * 4));
        SCS_TRY(d_rtiles.alloc(recv_tiles.size() * sizeof(int2)));
        ev_pair ev_x;
        SCS_TRY(ev_x.init());
        SCS_HIP_CHECK(hipEventRecord(ev_x.a, s));
        SCS_HIP_CHECK(hipMemcpyAsync(d_slots.p, send_slots.data(), send_slots.size() * 4,
                                     hipMemcpyHostToDevice, s));
        SCS_HIP_CHECK(hipMemcpyAsync(d_rtiles.p, recv_tiles.data(), recv_tiles.size() * sizeof(int2),
                                     hipMemcpyHostToDevice, s));
        if (!send_slots.empty())
            k_pack_tiles<<<(unsigned)send_slots.size(), 256, 0, s>>>(
                (const double *)d_tile_out.p, (const int32_t *)d_slots.p, cols_per_tile, (double *)d_send.p);
        SCS_HIP_CHECK(hipGetLastError());
        SCS_TRY(scs_comm_alltoallv_f64(&ctx->comm, (const double *)d_send.p, send_off.data(),
                                       (double *)d_recv.p, recv_off.data(), s));
        if (!recv_tiles.empty())
            k_unpack_received<<<(unsigned)recv_tiles.size(), cols_per_tile, 0, s>>>(
                (const double *)d_recv.p, (const int2 *)d_rtiles.p, n, row_begin, row_end, g->d_w,
                g->ld);
        SCS_HIP_CHECK(hipGetLastError());
        SCS_HIP_CHECK(hipEventRecord(ev_x.b, s));
        SCS_HIP_CHECK(hipEventSynchronize(ev_x.b));  // (also keeps the host vectors alive long enough)
        SCS_HIP_CHECK(hipEventElapsedTime(&exch_ms, ev_x.a, ev_x.b));
        exch_bytes = (double)(recv_off[world] - (recv_off[rank + 1] - recv_off[rank])) * 8.0;
    }
    SCS_HIP_CHECK(hipEventRecord(ev_total.b, s));
    SCS_HIP_CHECK(hipEventSynchronize(ev_total.b));
    SCS_TRY(scs_tables_finish(ctx, tb));  // the range check of the chunks that arrived meanwhile
    float total_ms = 0.f;
    SCS_HIP_CHECK(hipEventElapsedTime(&total_ms, ev_total.a, ev_total.b));
    if (n_batches == 1) {
        SCS_HIP_CHECK(hipEventElapsedTime(&prep_ms, ev_prep.a, ev_prep.b));
        SCS_HIP_CHECK(hipEventElapsedTime(&acc_ms, ev_acc.a, ev_acc.b));
        if (single_batch_spec) spec_ms = acc_ms;
    }

    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->n_taxa = n;
        stats->n_trees = M;
        stats->row_begin = row_begin;
        stats->row_end = row_end;
        stats->symmetric = shared ? 2 : (sym ? 1 : 0);
        stats->exchange_ms = exch_ms;
        stats->exchange_bytes = exch_bytes;
        stats->n_tiles = (int32_t)tiles.size();
        stats->n_batches = n_batches;
        stats->spec_batches = spec_batches;
        stats->tree_parallel_batches = tree_par ? n_batches : 0;
        stats->spec_trees = spec_trees;
        stats->listed_batches = listed_batches;
        stats->spec_ms = spec_ms;
        stats->cell_trees = (double)tiles.size() * SCS_TR * cols_per_tile * (double)M;
        stats->prep_ms = prep_ms;
        stats->accumulate_ms = acc_ms;
        stats->total_ms = total_ms;
        stats->bytes_w = 8.0 * rows * (double)n;
        stats->bytes_tables = 16.0 * (double)tb->n_leaves + 8.0 * M;
    }
    gd.g = nullptr;
    *out = g;
    return SCS_OK;
}

extern "C" int scs_graph_contract(scs_ctx *ctx, scs_graph *g, const int32_t *group_start,
                                  int32_t n_groups, scs_graph **out) {
    SCS_REQUIRE(ctx && g && group_start && out, "scs_graph_contract: null argument");
    if (g->mf) {
        scs_set_error("scs_graph_contract: a matrix-free graph has no matrix to contract");
        return SCS_EUNSUP;
    }
    if (g->upper) {
        scs_set_error("scs_graph_contract: an SCS_BUILD_UPPER graph holds no whole rows; build it "
                      "row-partitioned (SCS_BUILD_SHARED) when the node contracts");
        return SCS_EUNSUP;
    }
    SCS_REQUIRE(n_groups >= 1 && n_groups <= g->n, "scs_graph_contract: bad group count %d",
                n_groups);
    SCS_REQUIRE(group_start[0] == 0 && group_start[n_groups] == g->n,
                "scs_graph_contract: group_start must span [0, %d]", g->n);
    int g_begin = -1, g_end = -1;
    for (int i = 0; i < n_groups; ++i) {
        SCS_REQUIRE(group_start[i] < group_start[i + 1], "scs_graph_contract: empty group %d", i);
        if (group_start[i] == g->row_begin) g_begin = i;
        if (group_start[i + 1] == g->row_end) g_end = i + 1;
    }
    SCS_REQUIRE(g_begin >= 0 && g_end > g_begin,
                "scs_graph_contract: a group straddles this rank's row block [%d, %d)",
                g->row_begin, g->row_end);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    scs_graph *ng = nullptr;
    SCS_TRY(graph_alloc(ctx, n_groups, g_begin, g_end, ctx->stream, &ng));
    int32_t *d_gs = nullptr;
    hipError_t e = scs_dev_malloc(ctx, (void **)&d_gs, (size_t)(n_groups + 1) * 4);
    if (e != hipSuccess) {
        scs_graph_free(ctx, ng);
        scs_set_error("scs_graph_contract: hipMalloc failed: %s", hipGetErrorString(e));
        return SCS_ENOMEM;
    }
    hipMemcpyAsync(d_gs, group_start, (size_t)(n_groups + 1) * 4, hipMemcpyHostToDevice,
                   ctx->stream);
    dim3 grid((unsigned)((n_groups + 255) / 256), (unsigned)std::min(g_end - g_begin, 65535));
    k_contract<<<grid, 256, 0, ctx->stream>>>(g->d_w, g->ld, g->row_begin, d_gs, g_begin, g_end,
                                              n_groups, ng->d_w, ng->ld);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    scs_dev_free(d_gs);
    if (e != hipSuccess) {
        scs_graph_free(ctx, ng);
        scs_set_error("scs_graph_contract: kernel failed: %s", hipGetErrorString(e));
        return SCS_EHIP;
    }
    scs_graph_free(ctx, g);
    *out = ng;
    return SCS_OK;
}

extern "C" int scs_graph_download_rows(scs_ctx *ctx, const scs_graph *g, int32_t first, int32_t count,
                                       double *out);

extern "C" int scs_graph_download(scs_ctx *ctx, const scs_graph *g, double *out) {
    SCS_REQUIRE(ctx && g && out, "scs_graph_download: null argument");
    SCS_REQUIRE(!g->mf, "scs_graph_download: a matrix-free graph has no matrix");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const size_t rows = (size_t)(g->row_end - g->row_begin);
    if (g->upper) return scs_graph_download_rows(ctx, g, g->row_begin, (int32_t)rows, out);
    SCS_HIP_CHECK(hipMemcpy2D(out, (size_t)g->n * 8, g->d_w, (size_t)g->ld * 8, (size_t)g->n * 8,
                              rows, hipMemcpyDeviceToHost));
    return SCS_OK;
}

extern "C" int scs_graph_download_rows(scs_ctx *ctx, const scs_graph *g, int32_t first,
                                       int32_t count, double *out) {
    SCS_REQUIRE(ctx && g && out, "scs_graph_download_rows: null argument");
    SCS_REQUIRE(!g->mf, "scs_graph_download_rows: a matrix-free graph has no matrix");
    SCS_REQUIRE(count >= 1 && first >= g->row_begin && first + count <= g->row_end,
                "scs_graph_download_rows: rows [%d, %d) outside this rank's block [%d, %d)", first,
                first + count, g->row_begin, g->row_end);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    SCS_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (g->upper) {
        // stored: columns [col0, n); defined: from the row's 256-column diagonal tile on.  The
        // rest comes back as zeros.
        const size_t wcols = (size_t)(g->n - g->col0);
        for (int32_t i = 0; i < count; ++i) memset(out + (size_t)i * g->n, 0, (size_t)g->n * 8);
        SCS_HIP_CHECK(hipMemcpy2D(out + g->col0, (size_t)g->n * 8,
                                  g->d_w + (int64_t)(first - g->row_begin) * g->ld, (size_t)g->ld * 8,
                                  wcols * 8, (size_t)count, hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < count; ++i) {
            const int32_t c0 = (first + i) / 256 * 256;
            if (c0 > g->col0) memset(out + (size_t)i * g->n + g->col0, 0, (size_t)(c0 - g->col0) * 8);
        }
        return SCS_OK;
    }
    SCS_HIP_CHECK(hipMemcpy2D(out, (size_t)g->n * 8, g->d_w + (int64_t)(first - g->row_begin) * g->ld,
                              (size_t)g->ld * 8, (size_t)g->n * 8, (size_t)count,
                              hipMemcpyDeviceToHost));
    return SCS_OK;
}

// degrees of ALL vertices on every rank (local rows computed here, the rest
// gathered), plus 1/sqrt(d)
// The degrees in two halves: `begin` enqueues the kernels and the copy of the degrees into page-locked
// staging, `finish` waits and takes the host-side sums.  scs_fiedler puts its allocations and memsets
// between the two (the host works while k_degrees streams W); everything else calls both at once.
int scs_graph_prepare_degrees_begin(scs_ctx *ctx, scs_graph *g, bool want_w32) {
    if (g->deg_stage) return SCS_OK;
    const int n = g->n;
    const int rows = g->row_end - g->row_begin;
    if (g->mf && !g->have_deg) {
        // matrix-free graph: the degrees are W applied to the vector of ones
        SCS_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        if (!g->d_deg) SCS_TRY(scs_block_alloc(ctx, (size_t)n * 8, (void **)&g->d_deg));
        if (!g->d_dinv) SCS_TRY(scs_block_alloc(ctx, (size_t)n * 8, (void **)&g->d_dinv));
        SCS_TRY(scs_matfree_apply(ctx, g, nullptr, 0, 4, nullptr, s));  // (null operand: ones; column 0 -> d_deg)
        k_dinv<<<(n + 255) / 256, 256, 0, s>>>(g->d_deg, n, g->d_dinv);
        SCS_HIP_CHECK(hipGetLastError());
        SCS_TRY(scs_pinned_get(ctx, (size_t)n * 8, &g->deg_stage));
        SCS_HIP_CHECK(hipMemcpyAsync(g->deg_stage, g->d_deg, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        return SCS_OK;
    }
    // (round 6: a row-partitioned rank keeps the image of ITS rows, all columns -- what k_symm streams)
    const bool img_rows = ctx->comm.world > 1 && !g->upper;
    bool need_img = want_w32 && !g->have_w32 && !g->upper &&
                    (img_rows || (ctx->comm.world == 1 && g->row_begin == 0 && rows == n));
    if (g->have_deg && !need_img) return SCS_OK;
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // (no room for the image: the solve goes without it)
    if (need_img && !g->d_w32) {
        const size_t need = (size_t)rows * g->ld * 4;
        std::lock_guard<std::mutex> lock(ctx->cache_mu);
        if (ctx->w32_cache && ctx->w32_cache_bytes >= need) {
            g->d_w32 = ctx->w32_cache;
            g->w32_bytes = ctx->w32_cache_bytes;
            ctx->w32_cache = nullptr;
            ctx->w32_cache_bytes = 0;
        } else if (scs_dev_malloc(ctx, (void **)&g->d_w32, need) == hipSuccess) {
            g->w32_bytes = need;
            if (scs_dbg("SCS_TRACE_SOLVES") && atoi(scs_dbg("SCS_TRACE_SOLVES")))
                fprintf(stderr, "[image] hipMalloc of %.2f GB\n", need / 1073741824.0);
        } else {
            (void)hipGetLastError();
            g->d_w32 = nullptr;
            need_img = false;
        }
    }
    if (!need_img && g->have_deg) return SCS_OK;
    if (need_img && img_rows && g->have_deg) {
        // the degrees are known (a second solve on this graph): only the image is missing -- no collective here
        // (a rank whose image could not be allocated has returned above; it streams W in double precision)
        dev_buf tmp(ctx);
        SCS_TRY(tmp.alloc((size_t)n * 8));
        k_degrees<true><<<(rows + 3) / 4, 256, 0, s>>>(g->d_w, g->ld, n, rows, g->row_begin, (double *)tmp.p, g->d_w32, 1);
        SCS_HIP_CHECK(hipGetLastError());
        SCS_HIP_CHECK(hipStreamSynchronize(s));  // tmp goes out of scope
        g->have_w32 = true;
        return SCS_OK;
    }
    if (!g->d_deg) SCS_TRY(scs_block_alloc(ctx, (size_t)n * 8, (void **)&g->d_deg));
    if (!g->d_dinv) SCS_TRY(scs_block_alloc(ctx, (size_t)n * 8, (void **)&g->d_dinv));
    const int world = ctx->comm.world;
    if (need_img && !img_rows) {
        // (a graph whose degrees are known already gets them again, bit for bit, beside the image)
        k_degrees<true><<<(rows + 3) / 4, 256, 0, s>>>(g->d_w, g->ld, n, rows, g->row_begin, g->d_deg, g->d_w32);
        g->have_w32 = true;
    } else if (world == 1 && !g->upper) {
        k_degrees<false><<<(rows + 3) / 4, 256, 0, s>>>(g->d_w, g->ld, n, rows, g->row_begin, g->d_deg);
    } else {
        // every rank contributes a V-long vector that is zero outside its rows
        dev_buf send(ctx), recv(ctx);
        SCS_TRY(send.alloc((size_t)n * 8));
        SCS_TRY(recv.alloc((size_t)n * 8 * world));
        SCS_HIP_CHECK(hipMemsetAsync(send.p, 0, (size_t)n * 8, s));
        if (g->upper) {
            k_degrees_upper_rows<<<(rows + 3) / 4, 256, 0, s>>>(g->d_w, g->ld, n, rows, g->row_begin,
                                                                g->col0, (double *)send.p);
            k_degrees_upper_cols<<<(n - g->col0 + 255) / 256, 256, 0, s>>>(
                g->d_w, g->ld, n, g->row_begin, g->row_end, g->col0, (double *)send.p);
        } else if (need_img) {
            k_degrees<true><<<(rows + 3) / 4, 256, 0, s>>>(g->d_w, g->ld, n, rows, g->row_begin, (double *)send.p,
                                                           g->d_w32, 1);
            g->have_w32 = true;
        } else
        k_degrees<false><<<(rows + 3) / 4, 256, 0, s>>>(g->d_w, g->ld, n, rows, g->row_begin,
                                                        (double *)send.p);
        SCS_TRY(scs_comm_allgather_f64(&ctx->comm, (const double *)send.p, (double *)recv.p,
                                       (size_t)n, s));
        // every index is non-zero in its owner's slice only: the sum in rank order is exact
        k_combine_degrees<<<(n + 255) / 256, 256, 0, s>>>((const double *)recv.p, world, n, g->d_deg);
        SCS_HIP_CHECK(hipGetLastError());
        SCS_HIP_CHECK(hipStreamSynchronize(s));  // send / recv go out of scope
    }
    if (g->have_deg) {
        SCS_HIP_CHECK(hipGetLastError());
        return SCS_OK;
    }
    k_dinv<<<(n + 255) / 256, 256, 0, s>>>(g->d_deg, n, g->d_dinv);
    SCS_HIP_CHECK(hipGetLastError());
    SCS_TRY(scs_pinned_get(ctx, (size_t)n * 8, &g->deg_stage));
    SCS_HIP_CHECK(hipMemcpyAsync(g->deg_stage, g->d_deg, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    return SCS_OK;
}

int scs_graph_prepare_degrees(scs_ctx *ctx, scs_graph *g) {
    if (g->have_deg) return SCS_OK;
    SCS_TRY(scs_graph_prepare_degrees_begin(ctx, g));
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    const double *deg = (const double *)g->deg_stage;
    int iso = 0;
    double nrm2 = 0.0;
    if (e == hipSuccess)
        for (int i = 0; i < g->n; ++i) {
            if (deg[i] == 0.0) {
                ++iso;
                nrm2 += 1.0;
            } else {
                nrm2 += deg[i];
            }
        }
    scs_pinned_release(ctx, g->deg_stage);
    g->deg_stage = nullptr;
    SCS_HIP_CHECK(e);
    g->n_isolated = iso;
    g->dd_norm = sqrt(nrm2);
    g->have_deg = true;
    return SCS_OK;
}

extern "C" int scs_graph_degrees(scs_ctx *ctx, scs_graph *g, double *out) {
    SCS_REQUIRE(ctx && g && out, "scs_graph_degrees: null argument");
    SCS_TRY(scs_graph_prepare_degrees(ctx, g));
    SCS_HIP_CHECK(hipMemcpy(out, g->d_deg + g->row_begin,
                            (size_t)(g->row_end - g->row_begin) * 8, hipMemcpyDeviceToHost));
    return SCS_OK;
}
