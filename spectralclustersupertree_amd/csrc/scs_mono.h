// Monotone fast path of the proper-cluster-graph build (included by scs_build.hip).
//
// When the weighting value never decreases from an ancestor to a descendant (`one`, `depth`,
// `branch` with non-negative lengths and weights; host flag SCS_BUILD_MONOTONE), value(LCA)
// is a monotone image of the LCA depth: the value of the shallowest LCA of a range of gaps
// is the SMALLEST value in the range.  Everything can then be done on values:
//   * the per-tree range-minimum tables hold values (k_positions_values, k_sparse_level),
//   * the 64 x 64 row-row table of a (row block, tree) holds values,
//   * a cell is   acc += min(Dv[nb(c)][i], vn(c))   -- one ds_read_b64, one v_min_f64, one
//     v_add_f64 -- with nb(c) the tile row next to column c in DFS order whose LCA with c has
//     the larger value vn(c) (on a tie of VALUES between the two neighbours either gives the
//     same cells: all nodes between the two LCAs on c's root path then carry that value).
// Same addends in the same (tree) order as the general kernel and the reference
// (src/sc_supertree/scs.py:644-658), hence the same bits.
//
// Round-2 structure (profiles/r02_accumulate_*.txt: the round-1 kernel was LDS-bound, 73 %
// LDS-busy with half of it bank-conflict cycles, and re-expanded the same row-row table in
// each of the 20-200 column-group workgroups of a row block):
//   * ONE range-minimum query per column instead of two.  A column c falls between two tile
//     rows that are neighbours in DFS order; the smaller of its two LCA values equals the gap
//     value g between those rows, and it lies on the side that holds the position of g's
//     minimum, which the record carries (argpos): only the OTHER side is queried -- its value
//     is vn(c), its row nb(c).  With the tables over values there is no dependent value
//     gather either: 2 loads per (row block, tree, column) where round 1 had 5.  (Measured:
//     with the cell loop and the table staging switched off the kernel took as long as with
//     them -- it is bound by the 128-byte L2 lines these random 8-byte gathers move.)
//   * the column step of tree t+1 (search among the tile's rows + the two loads) is issued
//     one tree ahead and stays in flight across the barrier and the cell loop of tree t,
//   * the cell loop is hand-scheduled (scs_cells_asm.h),
//   * every address is a scalar base + 32-bit lane offset (buffer resources): the 64
//     accumulators leave no room for 64-bit per-lane addresses.
// (Staging a precomputed 33 KB table per step by LDS-DMA instead of expanding it was tried:
// +50 % -- it adds as much L2 traffic as the gathers it sits beside.)
// (Pacing the workgroups inside a launch was tried too: the workgroups (b >> 3) / 96 of one
// XCD met on a device-memory counter every 32-96 trees, bounded wait, so that one launch could
// take 256 trees of 50 000 leaves without drifting apart.  Slower at every interval -- 50 000
// leaves: 778 / 739 / 718 ms at 32 / 64 / 96 trees against 685 ms for plain launches of 80
// trees; 10 000 leaves: 8.6 against 7.2 ms.  The workgroups do not drift at random, they run at
// persistently different speeds (about +-20 % over 64 trees), and a meeting point makes every
// one of them as slow as the slowest.)
#pragma once

#include "scs_cells_asm.h"

constexpr int DV_LD = 65;  // leading dimension of the row-row table (doubles): a lane's
                           // ds_read_b64 of row nb hits bank pair (nb + i) mod 32
constexpr int DT_DOUBLES = 64 * DV_LD;
// per-(row block, tree) record of the monotone path
constexpr int R3_SPOS = 0;       // int32[64]  sorted DFS positions (INT_MAX beyond cnt)
constexpr int R3_G = 256;        // f64[64]    value of LCA(sorted k, sorted k+1); 0 beyond cnt-1
// The tile of the monotone kernel is 64 rows x MONO_TCW columns, one column per thread.  Wider
// tiles would share the table expansion (per tile and tree, independent of the width) among
// more columns, but measured slower: 384 columns (six waves) leave a CU with ONE workgroup --
// two six-wave workgroups need a (2,2,1,1) + (1,1,2,2) placement over the four SIMDs at three
// waves per SIMD, which the dispatcher does not find (+40 %); 768 columns (twelve waves, one
// workgroup per CU) serialise the step's phases behind its two barriers (+22 %).  Four waves,
// three workgroups per CU it is.  Wave w walks the expansion steps b in
// [mono_seg(w), mono_seg(w + 1)).
constexpr int MONO_WAVES = 4;
constexpr int MONO_TCW = 64 * MONO_WAVES;
__host__ __device__ constexpr int mono_seg(int w) { return 64 * w / MONO_WAVES; }
constexpr int MONO_MAXSEG = (64 + MONO_WAVES - 1) / MONO_WAVES;  // longest segment: 11 steps
constexpr int R3_SEED = 768;     // f64[5][64] seed[w-1][a] = min g[a .. mono_seg(w)-1] for a < mono_seg(w)
constexpr int R3_ARGPOS = R3_SEED + (MONO_WAVES - 1) * 512;  // int32[64] a position where gap k attains its minimum g[k]
constexpr int R3_SORIG = R3_ARGPOS + 256;  // u8[64]     row (0..63) at sorted rank k
constexpr int R3_RANK = R3_SORIG + 64;     // u8[64]     sorted rank of row i (absent rows: the ranks >= cnt)
constexpr int R3_PIV = R3_RANK + 64;       // int32[8]   sorted positions 7, 15, ..., 63 (search pivots)
constexpr int R3_CNT = R3_PIV + 32;        // int32      rows present in the tree
constexpr int R3_M = R3_CNT + 4;           // int32      gaps of the tree (n_t - 1)
constexpr int R3_STOFF = R3_M + 4;         // int64      offset of the tree's value table in the batch
constexpr int R3_PIECE = (R3_STOFF + 8 + MONO_WAVES * 16 - 1) / (MONO_WAVES * 16) * 16;  // bytes of a record one wave stages
constexpr int R3_BYTES = R3_PIECE * MONO_WAVES;
static_assert(R3_STOFF + 8 <= R3_BYTES && R3_PIECE <= 1024, "record layout");

// s_barrier without the fences of __syncthreads(): those make the compiler drain vmcnt in
// front of it, and with it the range-minimum loads that are meant to stay in flight across
// the barrier.  LDS traffic of this wave is drained here (lgkmcnt); the LDS-DMA data is
// ordered by the explicit vmcnt waits.
#define SCS_BARE_BARRIER()                                        \
    do {                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
        __builtin_amdgcn_s_barrier();                             \
        asm volatile("" ::: "memory");                            \
    } while (0)

// one wave per (local row block, tree): grid (n_blocks, trees in batch), 64 threads
__global__ __launch_bounds__(64) void k_block_records_mono(
    const int64_t *__restrict__ tree_off, int t0, int n_batch, const int32_t *__restrict__ pos,
    int64_t npad, const int64_t *__restrict__ st_off, const double *__restrict__ stv,
    int row_begin, int row_end, unsigned char *__restrict__ rec_all) {
    const int blk = blockIdx.x;
    const int tl = blockIdx.y;
    const int lane = threadIdx.x;
    const int m = (int)(tree_off[t0 + tl + 1] - tree_off[t0 + tl]) - 1;
    const int row = row_begin + blk * SCS_TR + lane;
    int p = -1;
    if (row < row_end) p = pos[(int64_t)tl * npad + row];
    const u32 pk = p < 0 ? 0x7FFFFFFFu : (u32)p;
    u64 key = ((u64)pk << 32) | (u32)lane;
    // bitonic sort of 64 unique keys across the wave
    for (int k = 2; k <= 64; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const u64 other = __shfl_xor(key, j, 64);
            const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
            const u64 lo = key < other ? key : other;
            const u64 hi = key < other ? other : key;
            key = take_min ? lo : hi;
        }
    }
    const int spos = (int)(key >> 32);
    const int orig = (int)(key & 63);
    const bool present = spos != 0x7FFFFFFF;
    const int cnt = __popcll(__ballot(present));
    const int next_pos = __shfl_down(spos, 1, 64);
    const double *st = stv + st_off[tl];
    // value of LCA(sorted rank k, sorted rank k + 1); 0 beyond the last present row, so every
    // table entry that involves an absent row comes out 0
    double g = 0.0;
    int argpos = 0;
    if (lane < cnt - 1) {
        g = rmq_min<double>(st, m, spos, next_pos);
        // a position in [spos, next_pos) where the minimum is attained: halve the range,
        // keeping a half whose minimum is still g.  (Eight sub-ranges a step -- seven prefix minima
        // in flight together, a third of the dependent round trips -- was measured: 587 us
        // against 450 for the 436-tree batch of 10 000 leaves: the kernel is bound by the number
        // of table gathers, not by their latency.)
        int lo = spos, hi = next_pos;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rmq_min<double>(st, m, lo, mid) == g) hi = mid;
            else lo = mid;
        }
        argpos = lo;
    }
    unsigned char *rec = rec_all + ((int64_t)blk * n_batch + tl) * R3_BYTES;
    ((int *)(rec + R3_SPOS))[lane] = spos;
    ((double *)(rec + R3_G))[lane] = g;
    ((int *)(rec + R3_ARGPOS))[lane] = argpos;
    if ((lane & 7) == 7) ((int *)(rec + R3_PIV))[lane >> 3] = spos;
    rec[R3_SORIG + lane] = (unsigned char)orig;
    rec[R3_RANK + orig] = (unsigned char)lane;
    if (lane == 0) {
        *(int *)(rec + R3_CNT) = cnt;
        *(int *)(rec + R3_M) = m;
        *(int64_t *)(rec + R3_STOFF) = st_off[tl];
    }
    // seeds of the table expansion: wave w of the tile kernel walks b from mono_seg(w) and needs,
    // for every lane a below that, the running minimum of g[a .. mono_seg(w) - 1]: a suffix
    // minimum cut off there, by doubling steps across the wave
    {
        const double inf = __longlong_as_double(0x7FF0000000000000ll);
        double *seed = (double *)(rec + R3_SEED);
#pragma unroll
        for (int w = 1; w < MONO_WAVES; ++w) {
            const int end = mono_seg(w);
            double mine = lane < end ? g : inf;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double other = __shfl_down(mine, off, 64);
                if (lane + off < end) mine = min_f64(mine, other);
            }
            seed[(w - 1) * 64 + lane] = mine;
        }
    }
}

struct mono_params {
    const int2 *tiles;           // (local row block, column group index)
    const unsigned char *rec;    // records, [block][tree]
    const int32_t *pos;          // [tree][npad] DFS position of a taxon, -1 if absent
    int64_t npad;
    const double *stv;           // value range-minimum tables of the batch
    int n_batch;
    double *w;                   // this rank's rows
    int64_t ld;
    int n;                       // V
    int row_begin, row_end;
    int load_w;                  // 1: continue a sum started by an earlier batch
    int mirror;                  // 1: last batch of a symmetric build: write the mirror image too
    double *tile_out;            // shared multi-rank build: packed 64 x 256 tiles (else null)
    unsigned long long *stamps;  // diagnostic build only (SCS_ACC_STAMP), else null
    int split_tiles;             // > 0: tree-parallel build of a small node (k_sum_tree_tiles): the
                                 // grid is split_tiles x n_batch, workgroup x takes tile x % split_tiles
                                 // of tree x / split_tiles ALONE and leaves its cells in tile_out slot x
    // partial-coverage forests (round 5, k_accumulate_mono<.., .., true>): tile i walks only the
    // trees lists[i * n_batch + 0 .. list_cnt[i]) -- those with a present row in its row block AND a
    // present column in its column group, in tree order (the others would add +0.0 to every cell)
    const int32_t *lists;
    const int32_t *list_cnt;
};

// ---- end of a tile (shared by the monotone and the general tile kernel): write the sums once,
// packed (shared multi-rank build) or straight into W, plus the mirror image in the last batch
// of a symmetric build.  s_dv: the workgroup's table space (free once the last tree is done).
template <bool SYM, typename P>
__device__ __forceinline__ void tile_store(const P &p, double (&acc)[SCS_TR], const int2 tile,
                                           const int row0, const int col, const int self,
                                           const int tid, const int lane, const int wave,
                                           double *s_dv) {
    if (self >= 0) {
#pragma unroll
        for (int i = 0; i < SCS_TR; ++i)
            if (i == self) acc[i] = 0.0;
    }
    if (p.tile_out) {
        double *tp = p.tile_out + (int64_t)blockIdx.x * SCS_TR * MONO_TCW + tid;
        if (p.split_tiles > 0) {
            // tree-parallel build: k_sum_tree_tiles reads the cells inside the matrix only -- a node of
            // 200 taxa fills 61 % of its four tiles
            if (col < p.n) {
#pragma unroll
                for (int i = 0; i < SCS_TR; ++i)
                    if (row0 + i < p.row_end) tp[i * MONO_TCW] = acc[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < SCS_TR; ++i) tp[i * MONO_TCW] = acc[i];
        return;
    }
    if (col < p.n) {
#pragma unroll
        for (int i = 0; i < SCS_TR; ++i) {
            const int r = row0 + i;
            if (r < p.row_end) p.w[(int64_t)(r - p.row_begin) * p.ld + col] = acc[i];
        }
    }
    if (SYM && p.mirror) {
        // The mirror image W[c][r] of the tile (cells no tile of the schedule owns; only the last
        // batch writes it -- earlier batches are re-read through the direct cells).  Stored
        // straight from the accumulators every lane would write 8 bytes into a different row;
        // instead eight rows at a time go through LDS ([column][8 rows], the table's space) and
        // come out as 64-byte runs along the rows of W.
        // (in the table's space when it fits, as it does at 256 columns)
        __shared__ double t_own[MONO_TCW * 9 > DT_DOUBLES ? MONO_TCW * 9 : 1];
        double *t = MONO_TCW * 9 > DT_DOUBLES ? t_own : s_dv;
        const bool wave_mirrors = ((row0 / MONO_TCW) + 1) * MONO_TCW <= ((tile.y * MONO_TCW + wave * 64) / SCS_TR) * SCS_TR;
#pragma unroll
        for (int q = 0; q < SCS_TR / 8; ++q) {
            SCS_BARE_BARRIER();
#pragma unroll
            for (int j = 0; j < 8; ++j) t[tid * 9 + j] = acc[q * 8 + j];
            SCS_BARE_BARRIER();
            if (wave_mirrors) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int cl = wave * 64 + it * 8 + (lane >> 3);  // column within the tile
                    const int c = tile.y * MONO_TCW + cl;
                    const int r = row0 + q * 8 + (lane & 7);
                    if (c < p.n && r < p.row_end) p.w[(int64_t)c * p.ld + r] = t[cl * 9 + (lane & 7)];
                }
            }
        }
    }
}

template <bool SYM, bool STAMPED, bool LISTED = false>
__global__ __launch_bounds__(MONO_TCW, MONO_TCW / 256) void k_accumulate_mono(mono_params p) {
    __shared__ __attribute__((aligned(16))) double s_dv[DT_DOUBLES];
    __shared__ __attribute__((aligned(16))) unsigned char s_rec[2][R3_BYTES];
    typedef __attribute__((address_space(3))) void *lds_ptr;

    // phase timers of the STAMPED diagnostic build (SCS_ACC_STAMP=1; never timed)
    unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if (STAMPED) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            ts[k] += tnow - tprev;
            tprev = tnow;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // (the wave index and everything read from the record are wave-uniform: saying so keeps
    // them in scalar registers)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = p.split_tiles;
    const int2 tile = p.tiles[split > 0 ? (int)blockIdx.x % split : (int)blockIdx.x];
    const int blk = tile.x;
    const int row0 = p.row_begin + blk * SCS_TR;
    const int col = tile.y * MONO_TCW + tid;
    const int nt = p.n_batch;
    // the trees this workgroup walks: all of the batch, or (tree-parallel build) one
    // (LISTED: positions in the tile's own list of trees; every use of a tree index below goes
    // through T(k), the parity of the record / table buffers follows k)
    typedef const __attribute__((address_space(4))) int32_t *clist;
    const clist lst = LISTED ? (clist)(size_t)(p.lists + (int64_t)blockIdx.x * nt) : nullptr;
    const int nl = LISTED ? __builtin_amdgcn_readfirstlane(p.list_cnt[blockIdx.x]) : nt;
    auto T = [&](int k) -> int { return LISTED ? lst[min(k, max(nl - 1, 0))] : k; };
    const int tl0 = LISTED ? 0 : (split > 0 ? (int)blockIdx.x / split : 0);
    const int tl1 = LISTED ? nl : (split > 0 ? tl0 + 1 : nt);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    // a column that is one of the tile's own rows (tiles on the diagonal): its cells are the
    // row-row table itself, no search or range-minimum needed; W[c][c] stays 0
    const int self = (col >= row0 && col < row0 + SCS_TR && col < p.row_end) ? col - row0 : -1;

    double acc[SCS_TR];
#pragma unroll
    for (int i = 0; i < SCS_TR; ++i) {
        double v = 0.0;
        if (p.load_w && p.tile_out)
            v = p.tile_out[((int64_t)blockIdx.x * SCS_TR + i) * MONO_TCW + tid];
        else if (p.load_w && col < p.n && row0 + i < p.row_end)
            v = p.w[(int64_t)(row0 - p.row_begin + i) * p.ld + col];
        acc[i] = v;
    }

    const unsigned char *rec_base = p.rec + (int64_t)blk * nt * R3_BYTES;
    const __amdgpu_buffer_rsrc_t r_rec =
        __builtin_amdgcn_make_buffer_rsrc((void *)rec_base, 0, nt * R3_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_pos =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.pos, 0, (int)(nt * p.npad * 4), 0x00020000);
    const int lane16 = lane * 16;
    const int col4 = col * 4;
    // every wave issues the same number of vector-memory operations per step (1 record piece,
    // 2 table loads, 1 position), so the counted vmcnt waits below hold for all of them;
    // wave w copies bytes [640 w, 640 w + 640) of a record with 40 lanes
    auto issue_record = [&](int tl, int b) {
        if (lane < R3_PIECE / 16)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rec, (lds_ptr)(s_rec[b] + wave * R3_PIECE), 16,
                                                     lane16, tl * R3_BYTES + wave * R3_PIECE, 0, 0);
    };

    // ---- column state of the tree whose range-minimum loads are in flight
    double qx = 0.0, qy = 0.0;  // raw table values, combined a step later
    int cstate = 0;             // bits 0-7: row nb; bit 8: a neighbour exists; bit 9: self
    int cpos_next = -1;

    // search tree tl's record (in s_rec[tl & 1]) for the column's position `cpos`, decide
    // which neighbour carries the larger LCA value and ISSUE the two loads of that ONE
    // range-minimum query; then request the position of the column in the next tree
    auto column_issue = [&](int tl, int cpos, int t_next) {
        const unsigned char *rb = s_rec[tl & 1];
        const int *s_spos = (const int *)(rb + R3_SPOS);
        const int *s_arg = (const int *)(rb + R3_ARGPOS);
        const unsigned char *s_sorig = rb + R3_SORIG;
        const int *s_piv = (const int *)(rb + R3_PIV);
        const int cnt = __builtin_amdgcn_readfirstlane(*(const int *)(rb + R3_CNT));
        const bool present = cpos >= 0 && cnt > 0;
        int lo;
        {
            // count of tile rows before the column: eight pivots (every eighth sorted
            // position, two independent 16-byte reads) pick the octet, three dependent
            // reads finish the count
            const int4 pa = *(const int4 *)&s_piv[0];
            const int4 pb = *(const int4 *)&s_piv[4];
            lo = ((pa.x < cpos) + (pa.y < cpos) + (pa.z < cpos) + (pa.w < cpos) + (pb.x < cpos) +
                  (pb.y < cpos) + (pb.z < cpos) + (pb.w < cpos)) * 8;
            const int base = min(lo, 56);  // lo == 64: all rows precede; reads stay in range
            int l2 = base;
#pragma unroll
            for (int s = 4; s > 0; s >>= 1)
                if (s_spos[l2 + s - 1] < cpos) l2 += s;
            lo = lo == 64 ? 64 : l2;
        }
        const bool hasl = present && self < 0 && lo > 0;
        const bool hasr = present && self < 0 && lo < cnt;
        const int il = max(lo - 1, 0), ir = min(lo, 63);
        // Between two rows: the gap's minimum sits left of the column <=> the LEFT LCA is the
        // gap value (the smaller one) and the right neighbour carries the larger value.
        const bool left = hasl && (!hasr || s_arg[il] >= cpos);
        const int q_anchor = s_spos[left ? il : ir];
        const int nbrow = s_sorig[left ? il : ir];
        cstate = nbrow | ((hasl || hasr) ? 256 : 0) | ((present && self >= 0) ? 512 : 0);
        const int m = __builtin_amdgcn_readfirstlane(*(const int *)(rb + R3_M));
        const unsigned so_lo = __builtin_amdgcn_readfirstlane(*(const unsigned *)(rb + R3_STOFF));
        const unsigned so_hi = __builtin_amdgcn_readfirstlane(*(const unsigned *)(rb + R3_STOFF + 4));
        const unsigned char *st = (const unsigned char *)(p.stv + (((u64)so_hi << 32) | so_lo));
        // left: gaps [anchor, cpos); right: gaps [cpos, anchor); a column without neighbours
        // reads entry 0 of the tree's level 0 (always addressable) and ignores it
        const bool any = hasl || hasr;
        int o[2];
        rmq_offsets(m, any ? (left ? q_anchor : cpos) : 0, any ? (left ? cpos : q_anchor) : 1, o);
        // (a tree's table is < 4 GiB: 32-bit byte offsets from a scalar base)
        qx = *(const double *)(st + (unsigned)o[0] * 8u);
        qy = *(const double *)(st + (unsigned)o[1] * 8u);
        cpos_next = __builtin_amdgcn_raw_buffer_load_b32(r_pos, col4, t_next * (int)p.npad * 4, 0);
    };

    // expand tree tl's row-row value table into s_dv.  In rank space entry (a, b), a < b, is
    // the running minimum of the gap values g[a .. b-1] (0 as soon as one of the rows is
    // absent), so the lane that owns the row of rank a carries `cur` along b and every step is
    // one v_min_f64 and two stores: the entry and its mirror image, at the rows' ORIGINAL
    // indices.  Lane = original row index i (rank rho_i): both stores are then free of bank
    // conflicts -- the mirror store writes row so_b contiguously, the other one has stride
    // DV_LD over consecutive lanes (with lane = rank they conflicted three ways).  Wave w walks
    // b in [mono_seg(w), mono_seg(w + 1)); lanes whose rank lies in an earlier segment take their running
    // minimum at the segment start from the record's seeds.  A lane's first active step
    // (b == rho) stores cur = +inf on the diagonal -- min(inf, vn) = vn: the cell (nb, c)
    // itself -- and picks up g[rho].
    auto expand = [&](int tl) {
        const unsigned char *rb = s_rec[tl & 1];
        const int b0 = mono_seg(wave), b1 = mono_seg(wave + 1);
        const double g_rank = ((const double *)(rb + R3_G))[lane];  // lane b holds g[b]
        const int so_rank = rb[R3_SORIG + lane];                    // lane b holds the row of rank b
        const int rho = rb[R3_RANK + lane];
        double cur = inf;
        if (rho < b0) cur = ((const double *)(rb + R3_SEED))[(wave - 1) * 64 + rho];
        double *row_a = &s_dv[lane * DV_LD];
        double *col_a = &s_dv[lane];
#pragma unroll
        for (int j = 0; j < MONO_MAXSEG; ++j) {
            const int b = b0 + j;
            if (b >= b1) break;  // uniform over the wave: segments are 10 or 11 steps long
            const int lo32 = __builtin_amdgcn_readlane((int)__double2loint(g_rank), b);
            const int hi32 = __builtin_amdgcn_readlane(__double2hiint(g_rank), b);
            const int so_b = __builtin_amdgcn_readlane(so_rank, b);
            const double gb = __hiloint2double(hi32, lo32);
            if (rho <= b) {
                row_a[so_b] = cur;
                col_a[so_b * DV_LD] = cur;
                cur = min_f64(cur, gb);
            }
        }
    };

    // ---- prologue: the records of the first two trees, the column's position in the first one;
    // then its column step
    if (LISTED && nl == 0) {  // (uniform: no tree of the batch touches this tile)
        tile_store<SYM>(p, acc, tile, row0, col, self, tid, lane, wave, s_dv);
        return;
    }
    issue_record(T(tl0), tl0 & 1);
    issue_record(T(min(tl0 + 1, nl - 1)), (tl0 + 1) & 1);
    cpos_next = __builtin_amdgcn_raw_buffer_load_b32(r_pos, col4, T(tl0) * (int)p.npad * 4, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    column_issue(tl0, cpos_next, T(min(tl0 + 1, nl - 1)));  // 3 operations in flight: 2 table loads + the next position
    if (STAMPED) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");

    // Order inside a step (tree tl).  The compiler cannot tell an LDS-DMA in flight from the
    // LDS it is about to touch and drains vmcnt in front of every ds_read / ds_write that
    // follows one, so the step's only DMA -- the record of tree tl + 2 -- is issued after the
    // last LDS access the compiler sees (the cell loop is opaque to it); the table loads of
    // the column step have the whole cell loop to come back.
    for (int tl = tl0; tl < tl1; ++tl) {
        // the column's loads for tree tl, its position in tree tl + 1, this wave's piece of
        // the record of tree tl + 1 (all issued a cell loop ago)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(0);
        // A: every wave is done with the cells of tree tl - 1 (s_dv); the record of tree
        // tl + 1 is complete
        SCS_BARE_BARRIER();
        stamp(1);
        // ---- combine tree tl's loads
        int nb = 0;
        double vn = 0.0;
        if (cstate & 256) {
            vn = min_f64(qx, qy);
            nb = cstate & 255;
        } else if (cstate & 512) {
            // cell (i, c) = table entry (self, i); the diagonal entry is +inf and makes
            // acc[self] meaningless -- it is reset after the last tree
            nb = self;
            vn = inf;
        }
        stamp(2);
        // (the last step searches its own tree again, result unused: keeps the step free of
        // branches the compiler would have to merge wait states over)
        column_issue(min(tl + 1, nl - 1), cpos_next, T(min(tl + 2, nl - 1)));  // 3 operations
        stamp(3);
        expand(tl);
        stamp(4);
        SCS_BARE_BARRIER();  // B: the table is complete; the record of tree tl is free
        stamp(5);
        issue_record(T(min(tl + 2, nl - 1)), tl & 1);
        {
            double tmp[SCS_CELLS_DEPTH];
            const unsigned addr =
                (unsigned)(size_t)(__attribute__((address_space(3))) double *)&s_dv[nb * DV_LD];
            SCS_CELLS_ASM(acc, tmp, addr, vn);
        }
        stamp(6);
    }

    if (STAMPED && lane == 0 && p.stamps) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&p.stamps[k], ts[k]);
        atomicAdd(&p.stamps[7], 1ull);
    }
    tile_store<SYM>(p, acc, tile, row0, col, self, tid, lane, wave, s_dv);
}

// ---- tree-parallel build of a small node (mid-size recursion nodes)
// A node of a few hundred taxa is a handful of tiles, and a tile's workgroup walks the batch's
// trees one after the other -- ~2 us a tree whatever the tile holds (barriers, the gathers of one
// range-minimum query, the table expansion): 10 ms for 5 000 trees with 250 CUs idle.  The sum
// has to be formed in tree order, but the ADDENDS need no order: k_accumulate_mono / _gen with
// p.split_tiles > 0 gives every (tile, tree) pair a workgroup of its own that leaves the tree's
// cells (0.0 + v = v: exact) in `cells` [tree][tile][64][MONO_TCW], and this kernel adds them up:
// one thread per cell, the trees in order, sixteen loads in flight -- the same additions in the
// same order as the walk, the same bits.  Pays while the cells' round trip through HBM (2 x 128 KB
// per tile and tree) is cheaper than the walk's step: up to ~12 tiles (scs_pcg_build decides).
template <bool SYM, typename P>
__global__ __launch_bounds__(MONO_TCW) void k_sum_tree_tiles(P p, const double *__restrict__ cells, int n_tiles) {
    const int tile_i = (int)blockIdx.x / SCS_TR, i = (int)blockIdx.x % SCS_TR;
    const int2 tile = p.tiles[tile_i];
    const int row = p.row_begin + tile.x * SCS_TR + i;
    const int col = tile.y * MONO_TCW + (int)threadIdx.x;
    if (row >= p.row_end || col >= p.n) return;
    double *wcell = p.w + (int64_t)(row - p.row_begin) * p.ld + col;
    double acc = p.load_w ? *wcell : 0.0;
    const double *src = cells + ((int64_t)tile_i * SCS_TR + i) * MONO_TCW + threadIdx.x;
    const int64_t stride = (int64_t)n_tiles * SCS_TR * MONO_TCW;
    int t = 0;
    for (; t + 16 <= p.n_batch; t += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[(int64_t)(t + k) * stride];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    for (; t < p.n_batch; ++t) acc += src[(int64_t)t * stride];
    *wcell = acc;
    // the mirror image of the cells no tile of the schedule owns (tile_store's rule), last batch
    if (SYM && p.mirror && (row / MONO_TCW + 1) * MONO_TCW <= (col / SCS_TR) * SCS_TR)
        p.w[(int64_t)col * p.ld + row] = acc;
}
