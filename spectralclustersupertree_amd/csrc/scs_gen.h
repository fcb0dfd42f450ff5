// General path of the proper-cluster-graph build on the tile structure of scs_mono.h
// (included by scs_build.hip after it): weighting values that are NOT monotone in the depth
// (`bootstrap`, negative branch lengths), where value(LCA) cannot be found by a minimum over
// values.  Everything the monotone kernel does on values is done here on (depth, value) pairs:
//   * the per-tree range-minimum tables hold 16-byte entries {depth, value * w} ordered by depth
//     (two gaps of one depth inside a range that has no shallower gap are the same node, so a
//     tie may be broken either way); a query is still two independent loads, no dependent value
//     gather,
//   * the record of a (row block, tree) carries depth and value of the 63 gaps between the
//     tile's rows in DFS order, the position of each gap's minimum (argpos), seeds of the table
//     expansion and a 6-level sparse table over the gap depths,
//   * ONE query per column, chosen as in the monotone kernel: the neighbour row nb(c) on the side
//     that does not hold the gap's minimum has the deeper LCA with the column -- depth dn(c),
//     value vv(c),
//   * a cell is  acc += depth(LCA(nb, i)) >= dn(c) ? vv(c) : value(LCA(nb, i)).  In DFS order the
//     rows with depth(LCA(nb, i)) >= dn(c) are a run of sorted ranks that starts at nb and extends
//     AWAY from the column (towards the column the very next row already has the shallower LCA),
//     so the run is found once per (column, tree) -- a 6-step descent over the record's sparse
//     table -- and the cell loop only compares the row's rank, wave-uniform, with the run's
//     ends: the 64 x 64 table holds values only, one ds_read_b64 per cell as in the monotone
//     kernel (scs_cells_asm.h, SCS_CELLS_GEN_ASM).
// Same addends in the same (tree) order as the reference (src/sc_supertree/scs.py:644-658):
// the same bits.  Absent rows and the gaps behind the last present row carry depth 0 and value
// 0, like a root LCA, which contributes 0 as well.
#pragma once

// 16 bytes, of which the first 12 carry data (the tile kernel loads just those: 3 registers)
struct __attribute__((aligned(16))) gap_entry {
    u32 d;        // depth of the gap's LCA (0: the root)
    u32 vlo, vhi; // value * tree weight (0 at depth 0), the two halves of the double
    u32 pad;
    __device__ __forceinline__ double v() const { return __hiloint2double((int)vhi, (int)vlo); }
    __device__ __forceinline__ void set_v(double x) {
        vlo = (u32)__double2loint(x);
        vhi = (u32)__double2hiint(x);
    }
};
typedef u32 gap_words __attribute__((ext_vector_type(3)));

__device__ __forceinline__ gap_entry gap_min(const gap_entry a, const gap_entry b) {
    return b.d < a.d ? b : a;
}

__device__ __forceinline__ gap_entry rmq_gap(const gap_entry *__restrict__ base, int m, int a, int b) {
    int o[2];
    rmq_offsets(m, a, b, o);
    return gap_min(base[o[0]], base[o[1]]);
}

// record of a (row block, tree)
constexpr int G3_SPOS = 0;                   // int32[64]  sorted DFS positions (INT_MAX beyond cnt)
constexpr int G3_GV = 256;                   // f64[64]    value of LCA(sorted k, sorted k+1); 0 beyond cnt-1
constexpr int G3_GD = 768;                   // u32[6][64] level j: min depth of gaps [k, k + 2^j) (level 0: the gaps)
constexpr int G3_SEEDV = G3_GD + 6 * 256;    // f64[3][64] value / depth of the shallowest gap in
constexpr int G3_SEEDD = G3_SEEDV + (MONO_WAVES - 1) * 512;  // u32[3][64]   [a, mono_seg(w)) for a < mono_seg(w)
constexpr int G3_ARGPOS = G3_SEEDD + (MONO_WAVES - 1) * 256;  // int32[64] a position where gap k is shallowest
constexpr int G3_SORIG = G3_ARGPOS + 256;    // u8[64]     row (0..63) at sorted rank k
constexpr int G3_RANK = G3_SORIG + 64;       // u8[64]     sorted rank of row i (absent rows: the ranks >= cnt)
constexpr int G3_PIV = G3_RANK + 64;         // int32[8]   sorted positions 7, 15, ..., 63 (search pivots)
constexpr int G3_CNT = G3_PIV + 32;          // int32      rows present in the tree
constexpr int G3_M = G3_CNT + 4;             // int32      gaps of the tree (n_t - 1)
constexpr int G3_STOFF = G3_M + 4;           // int64      offset (entries) of the tree's table in the batch
constexpr int G3_PIECE = (G3_STOFF + 8 + MONO_WAVES * 16 - 1) / (MONO_WAVES * 16) * 16;
constexpr int G3_BYTES = G3_PIECE * MONO_WAVES;
static_assert(G3_PIECE > 1024 && G3_PIECE <= 2048, "a wave stages its piece of a record with two DMAs");

// grid (ceil(max_leaves/256), trees in batch): positions and level 0 of the pair tables
__global__ void k_positions_pairs(const int64_t *__restrict__ tree_off,
                                  const int32_t *__restrict__ leaf_taxon,
                                  const int32_t *__restrict__ adj_depth,
                                  const double *__restrict__ adj_val,
                                  const double *__restrict__ tree_w, int t0,
                                  int32_t *__restrict__ pos, int64_t npad,
                                  const int64_t *__restrict__ st_off, gap_entry *__restrict__ ste) {
    const int tl = blockIdx.y;
    const int t = t0 + tl;
    const int64_t off = tree_off[t];
    const int n = (int)(tree_off[t + 1] - off);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    // (an id of a late table chunk may not have been range-checked yet: see k_positions_values)
    const unsigned tx = (unsigned)leaf_taxon[off + p];
    if (tx < (unsigned)npad) pos[(int64_t)tl * npad + tx] = p;
    if (p < n - 1) {
        gap_entry e;
        e.d = (u32)adj_depth[off + p];
        // one rounded multiply, as the reference's `length * tree_weight`
        e.set_v(e.d ? adj_val[off + p] * tree_w[t] : 0.0);
        e.pad = 0;
        ste[st_off[tl] + p] = e;
    }
}

// level k >= 1 of every tree's pair table; grid as k_positions_pairs
__global__ void k_sparse_level_pairs(const int64_t *__restrict__ tree_off, int t0, int k,
                                     const int64_t *__restrict__ st_off,
                                     gap_entry *__restrict__ ste) {
    const int tl = blockIdx.y;
    const int m = (int)(tree_off[t0 + tl + 1] - tree_off[t0 + tl]) - 1;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < (1 << k) || p > m - (1 << k)) return;
    gap_entry *base = ste + st_off[tl];
    base[(int64_t)k * m + p] =
        gap_min(base[(int64_t)(k - 1) * m + p], base[(int64_t)(k - 1) * m + p + (1 << (k - 1))]);
}

// one wave per (local row block, tree): grid (n_blocks, trees in batch), 64 threads
__global__ __launch_bounds__(64) void k_block_records_gen(
    const int64_t *__restrict__ tree_off, int t0, int n_batch, const int32_t *__restrict__ pos,
    int64_t npad, const int64_t *__restrict__ st_off, const gap_entry *__restrict__ ste,
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
    const gap_entry *st = ste + st_off[tl];
    u32 gd = 0;
    double gv = 0.0;
    int argpos = 0;
    if (lane < cnt - 1) {
        const gap_entry g = rmq_gap(st, m, spos, next_pos);
        gd = g.d;
        gv = g.v();
        // a position in [spos, next_pos) where the depth is smallest: halve the range, keeping
        // a half whose minimum is still gd
        int lo = spos, hi = next_pos;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rmq_gap(st, m, lo, mid).d == gd) hi = mid;
            else lo = mid;
        }
        argpos = lo;
    }
    unsigned char *rec = rec_all + ((int64_t)blk * n_batch + tl) * G3_BYTES;
    ((int *)(rec + G3_SPOS))[lane] = spos;
    ((double *)(rec + G3_GV))[lane] = gv;
    ((int *)(rec + G3_ARGPOS))[lane] = argpos;
    if ((lane & 7) == 7) ((int *)(rec + G3_PIV))[lane >> 3] = spos;
    rec[G3_SORIG + lane] = (unsigned char)orig;
    rec[G3_RANK + orig] = (unsigned char)lane;
    if (lane == 0) {
        *(int *)(rec + G3_CNT) = cnt;
        *(int *)(rec + G3_M) = m;
        *(int64_t *)(rec + G3_STOFF) = st_off[tl];
    }
    // sparse table over the gap depths (level j, entry k: gaps [k, k + 2^j); entries that would
    // reach past gap 62 are never read)
    {
        u32 *sp = (u32 *)(rec + G3_GD);
        u32 mine = gd;
        sp[lane] = mine;
#pragma unroll
        for (int j = 1; j < 6; ++j) {
            const u32 other = __shfl_down(mine, 1 << (j - 1), 64);
            if (lane + (1 << (j - 1)) < 64) mine = other < mine ? other : mine;
            sp[j * 64 + lane] = mine;
        }
    }
    // seeds of the table expansion: wave w of the tile kernel walks b from mono_seg(w) and needs,
    // for every lane a below that, the shallowest gap of [a, mono_seg(w)): a suffix minimum cut
    // off there, by doubling steps across the wave (ties: any -- the same node)
    {
        double *seedv = (double *)(rec + G3_SEEDV);
        u32 *seedd = (u32 *)(rec + G3_SEEDD);
#pragma unroll
        for (int w = 1; w < MONO_WAVES; ++w) {
            const int end = mono_seg(w);
            u32 md = lane < end ? gd : DEPTH_INF;
            double mv = gv;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 od = __shfl_down(md, off, 64);
                const double ov = __shfl_down(mv, off, 64);
                if (lane + off < end && od < md) {
                    md = od;
                    mv = ov;
                }
            }
            seedd[(w - 1) * 64 + lane] = md;
            seedv[(w - 1) * 64 + lane] = mv;
        }
    }
}

struct gen_params {
    const int2 *tiles;           // (local row block, column group index)
    const unsigned char *rec;    // records, [block][tree]
    const int32_t *pos;          // [tree][npad] DFS position of a taxon, -1 if absent
    int64_t npad;
    const gap_entry *ste;        // pair range-minimum tables of the batch
    int n_batch;
    double *w;                   // this rank's rows
    int64_t ld;
    int n;                       // V
    int row_begin, row_end;
    int load_w;                  // 1: continue a sum started by an earlier batch
    int mirror;                  // 1: last batch of a symmetric build: write the mirror image too
    double *tile_out;            // shared multi-rank build: packed 64 x 256 tiles (else null)
    int split_tiles;             // > 0: tree-parallel build (scs_mono.h, k_sum_tree_tiles)
};

template <bool SYM>
__global__ __launch_bounds__(MONO_TCW, 3) void k_accumulate_gen(gen_params p) {
    __shared__ __attribute__((aligned(16))) double s_dv[DT_DOUBLES];
    __shared__ __attribute__((aligned(16))) unsigned char s_rec[2][G3_BYTES];
    typedef __attribute__((address_space(3))) void *lds_ptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = p.split_tiles;
    const int2 tile = p.tiles[split > 0 ? (int)blockIdx.x % split : (int)blockIdx.x];
    const int blk = tile.x;
    const int row0 = p.row_begin + blk * SCS_TR;
    const int col = tile.y * MONO_TCW + tid;
    const int nt = p.n_batch;
    // the trees this workgroup walks: all of the batch, or (tree-parallel build) one
    const int tl0 = split > 0 ? (int)blockIdx.x / split : 0;
    const int tl1 = split > 0 ? tl0 + 1 : nt;
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

    const unsigned char *rec_base = p.rec + (int64_t)blk * nt * G3_BYTES;
    const __amdgpu_buffer_rsrc_t r_rec =
        __builtin_amdgcn_make_buffer_rsrc((void *)rec_base, 0, nt * G3_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_pos =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.pos, 0, (int)(nt * p.npad * 4), 0x00020000);
    const int lane16 = lane * 16;
    const int col4 = col * 4;
    // wave w copies its piece of a record: 1 KiB by all lanes, the rest by the first few
    auto issue_record = [&](int tl, int b) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rec, (lds_ptr)(s_rec[b] + wave * G3_PIECE), 16,
                                                 lane16, tl * G3_BYTES + wave * G3_PIECE, 0, 0);
        if (lane < (G3_PIECE - 1024) / 16)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rec, (lds_ptr)(s_rec[b] + wave * G3_PIECE + 1024),
                                                     16, lane16, tl * G3_BYTES + wave * G3_PIECE + 1024,
                                                     0, 0);
    };

    // ---- column state of the tree whose range-minimum loads are in flight
    gap_words qx = {0, 0, 0}, qy = {0, 0, 0};  // raw table entries (depth, value), combined a step later
    // bits 0-7: row nb; bit 8: a neighbour exists; bit 9: self; bit 10: nb is the LEFT
    // neighbour; bits 16-21: nb's sorted rank
    int cstate = 0;
    int cpos_next = -1;

    // search tree tl's record (in s_rec[tl & 1]) for the column's position `cpos`, decide
    // which neighbour carries the deeper LCA and ISSUE the two loads of that ONE range-minimum
    // query; then request the position of the column in the next tree
    auto column_issue = [&](int tl, int cpos) {
        const unsigned char *rb = s_rec[tl & 1];
        const int *s_spos = (const int *)(rb + G3_SPOS);
        const int *s_arg = (const int *)(rb + G3_ARGPOS);
        const unsigned char *s_sorig = rb + G3_SORIG;
        const int *s_piv = (const int *)(rb + G3_PIV);
        const int cnt = __builtin_amdgcn_readfirstlane(*(const int *)(rb + G3_CNT));
        const bool present = cpos >= 0 && cnt > 0;
        int lo;
        {
            // count of tile rows before the column: eight pivots pick the octet, three
            // dependent reads finish the count
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
        // Between two rows: the gap's shallowest point sits left of the column <=> the LEFT LCA
        // is the gap's (the shallower one) and the right neighbour carries the deeper LCA.
        const bool left = hasl && (!hasr || s_arg[il] >= cpos);
        const int ra = left ? il : ir;
        const int q_anchor = s_spos[ra];
        const int nbrow = s_sorig[ra];
        cstate = nbrow | ((hasl || hasr) ? 256 : 0) | ((present && self >= 0) ? 512 : 0) |
                 (left ? 1024 : 0) | (ra << 16);
        const int m = __builtin_amdgcn_readfirstlane(*(const int *)(rb + G3_M));
        const unsigned so_lo = __builtin_amdgcn_readfirstlane(*(const unsigned *)(rb + G3_STOFF));
        const unsigned so_hi = __builtin_amdgcn_readfirstlane(*(const unsigned *)(rb + G3_STOFF + 4));
        const unsigned char *st = (const unsigned char *)(p.ste + (((u64)so_hi << 32) | so_lo));
        // left: gaps [anchor, cpos); right: gaps [cpos, anchor); a column without neighbours
        // reads entry 0 of the tree's level 0 (always addressable) and ignores it
        const bool any = hasl || hasr;
        int o[2];
        rmq_offsets(m, any ? (left ? q_anchor : cpos) : 0, any ? (left ? cpos : q_anchor) : 1, o);
        // (a tree's table is < 4 GiB: 32-bit byte offsets from a scalar base)
        qx = *(const gap_words *)(st + (unsigned)o[0] * 16u);
        qy = *(const gap_words *)(st + (unsigned)o[1] * 16u);
        cpos_next = __builtin_amdgcn_raw_buffer_load_b32(
            r_pos, col4, min(tl + 1, nt - 1) * (int)p.npad * 4, 0);
    };

    // expand tree tl's row-row value table into s_dv.  In rank space entry (a, b), a < b, is the
    // value of the shallowest of the gaps a .. b-1, so the lane that owns the row of rank a
    // carries (depth, value) of the shallowest gap so far along b; every step stores the value
    // twice -- the entry and its mirror image, at the rows' ORIGINAL indices (lane = original
    // row index: both stores free of bank conflicts, see scs_mono.h).  Wave w walks b in
    // [mono_seg(w), mono_seg(w + 1)); lanes whose rank lies in an earlier segment start from the
    // record's seeds.  The diagonal entry is never used for a column with a neighbour (its own
    // rank is inside its run) and is reset for a self column.
    auto expand = [&](int tl) {
        const unsigned char *rb = s_rec[tl & 1];
        const int b0 = mono_seg(wave), b1 = mono_seg(wave + 1);
        const double gv_rank = ((const double *)(rb + G3_GV))[lane];  // lane b holds gap b
        const int gd_rank = ((const int *)(rb + G3_GD))[lane];
        const int so_rank = rb[G3_SORIG + lane];                      // lane b holds the row of rank b
        const int rho = rb[G3_RANK + lane];
        u32 curd = DEPTH_INF;
        double curv = 0.0;
        if (rho < b0) {
            curd = ((const u32 *)(rb + G3_SEEDD))[(wave - 1) * 64 + rho];
            curv = ((const double *)(rb + G3_SEEDV))[(wave - 1) * 64 + rho];
        }
        double *row_a = &s_dv[lane * DV_LD];
        double *col_a = &s_dv[lane];
#pragma unroll
        for (int j = 0; j < MONO_MAXSEG; ++j) {
            const int b = b0 + j;
            if (b >= b1) break;  // uniform over the wave
            const int lo32 = __builtin_amdgcn_readlane((int)__double2loint(gv_rank), b);
            const int hi32 = __builtin_amdgcn_readlane(__double2hiint(gv_rank), b);
            const u32 gdb = (u32)__builtin_amdgcn_readlane(gd_rank, b);
            const int so_b = __builtin_amdgcn_readlane(so_rank, b);
            const double gvb = __hiloint2double(hi32, lo32);
            if (rho <= b) {
                row_a[so_b] = curv;
                col_a[so_b * DV_LD] = curv;
                const bool take = gdb < curd;
                curd = take ? gdb : curd;
                curv = take ? gvb : curv;
            }
        }
    };

    // ---- prologue: records 0 and 1, the column's position in tree 0; then tree 0's column step
    issue_record(tl0, tl0 & 1);
    issue_record(min(tl0 + 1, nt - 1), (tl0 + 1) & 1);
    cpos_next = __builtin_amdgcn_raw_buffer_load_b32(r_pos, col4, tl0 * (int)p.npad * 4, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    column_issue(tl0, cpos_next);

    // Order inside a step as in k_accumulate_mono: the step's only LDS-DMA (the record of tree
    // tl + 2) is issued after the last LDS access the compiler sees.
    for (int tl = tl0; tl < tl1; ++tl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // A: every wave is done with the cells of tree tl - 1 (s_dv); the record of tree
        // tl + 1 is complete
        SCS_BARE_BARRIER();
        // ---- combine tree tl's loads: depth and value of the column's LCA with nb, then the
        // run of sorted ranks [tlo, tlo + tw] whose rows share that LCA with the column
        const unsigned char *rb = s_rec[tl & 1];
        int nb = 0;
        double vv = 0.0;
        unsigned tlo = 0, tw = 63;  // absent column: every cell takes vv = 0
        if (cstate & 256) {
            const gap_words g = qy.x < qx.x ? qy : qx;
            const u32 dn = g.x;
            vv = __hiloint2double((int)g.z, (int)g.y);
            nb = cstate & 255;
            const int ra = (cstate >> 16) & 63;
            const bool left = (cstate & 1024) != 0;
            const u32 *sp = (const u32 *)(rb + G3_GD);
            int pos = ra;
#pragma unroll
            for (int j = 5; j >= 0; --j) {
                const int step = 1 << j;
                const int idx = left ? pos - step : pos;
                const bool ok = left ? idx >= 0 : pos + step <= 63;
                const u32 dmin = sp[j * 64 + (ok ? idx : 0)];
                if (ok && dmin >= dn) pos = left ? idx : pos + step;
            }
            tlo = left ? pos : ra;
            tw = left ? ra - pos : pos - ra;
        } else if (cstate & 512) {
            // cell (i, c) = table entry (self, i) for every row; acc[self] is reset after the
            // last tree
            nb = self;
            tlo = 64;
            tw = 0;
        }
        // (the last step searches its own tree again, result unused)
        column_issue(min(tl + 1, nt - 1), cpos_next);
        expand(tl);
        // the ranks of the 64 rows, four to a dword, on their way to scalar registers
        const int rank_pack = ((const int *)(rb + G3_RANK))[lane & 15];
        SCS_BARE_BARRIER();  // B: the table is complete; the record of tree tl is free
        issue_record(min(tl + 2, nt - 1), tl & 1);
        {
            int rp[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) rp[k] = __builtin_amdgcn_readlane(rank_pack, k);
            double tmp[SCS_CELLS_GEN_DEPTH];
            unsigned tt;
            int sr;
            const unsigned addr =
                (unsigned)(size_t)(__attribute__((address_space(3))) double *)&s_dv[nb * DV_LD];
            SCS_CELLS_GEN_ASM(acc, tmp, tt, sr, addr, vv, tlo, tw, rp);
        }
    }

    tile_store<SYM>(p, acc, tile, row0, col, self, tid, lane, wave, s_dv);
}
