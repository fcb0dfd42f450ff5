// k_accumulate_spec: the monotone tile kernel with producer and consumer waves (included by
// scs_build.hip after scs_mono.h).  Round 4.
//
// profiles/r03_accumulate_phase_stamps.txt and the LDS cycle count of DESIGN.md 3.1: of the
// ~1 900 LDS cycles a (64 x 256 tile, tree) step of k_accumulate_mono costs, 768 are the stores
// that expand the tile's 64 x 64 row-row table -- and every one of the 40 (10 000 taxa) to 196
// (50 000) column tiles of a row block expands the SAME table again; and the LDS is busy two
// thirds of the time, because a wave's step is a chain of dependent phases (column search,
// range-minimum loads, table expansion, cells) that it walks alone.  Three kernels were built
// on the way here and measured against k_accumulate_mono, all bit-exact (profiles/
// r04_wide_phase_stamps.txt, r04_cells_probe.txt):
//   * twelve waves on three column tiles of one row block, two table buffers, one barrier per
//     tree: the table is expanded once for 768 columns, yet the step is no shorter -- the twelve
//     waves search, expand and read cells in step with each other and a quarter of the time goes
//     to the barrier (+10 %);
//   * eight waves on two tiles with the expansion WOVEN INTO the cell loop and the two tiles
//     half a step apart: the woven statement runs at exactly the LDS cycles of the step, the LDS
//     idles outside it (+24 %);
//   * a microbenchmark of the inner loops alone: twelve waves that do nothing but cells,
//     expansion stores and one barrier per tree keep the LDS 92 % busy (7.5e12 cell-trees/s); a
//     ROW-lane cell loop (conflict-free reads, scalar operands through s_load) is slower than
//     the column-lane one at every occupancy.
// What is left is this kernel: the roles are split (below).  The external tile geometry
// (64 x 256, tile lists, packed tiles of the shared multi-rank build, mirror image) is
// unchanged: a workgroup is handed two tiles of one row block (`groups`).  Arithmetic as
// k_accumulate_mono: same addends, same order, same bits (reference:
// src/sc_supertree/scs.py:644-658).
#pragma once

template <int NG>
struct wide_layout {
    static constexpr int NSEG = 4 * NG;  // expansion segments = waves
    static constexpr int SPOS = 0;       // int32[64]  sorted DFS positions (INT_MAX beyond cnt)
    static constexpr int G = 256;        // f64[64]    value of LCA(sorted k, sorted k+1); 0 beyond cnt-1
    static constexpr int SEED = 768;     // f64[NSEG-1][64] seed[w-1][a] = min g[a .. seg(w)-1] for a < seg(w)
    static constexpr int ARGPOS = SEED + (NSEG - 1) * 512;  // int32[64]
    static constexpr int SORIG = ARGPOS + 256;              // u8[64]
    static constexpr int RANK = SORIG + 64;                 // u8[64]
    static constexpr int PIV = RANK + 64;                   // int32[8]
    static constexpr int CNT = PIV + 32;                    // int32
    static constexpr int M = CNT + 4;                       // int32
    static constexpr int STOFF = M + 4;                     // int64
    static constexpr int BYTES = (STOFF + 8 + 15) / 16 * 16;
    static constexpr int PIECES = (BYTES + 1023) / 1024;    // 1 KiB pieces of a record, one or two per producer wave
    __host__ __device__ static constexpr int seg(int w) { return 64 * w / NSEG; }
    static constexpr int MAXSEG = (64 + NSEG - 1) / NSEG;
    static_assert(STOFF % 8 == 0 && PIECES <= NSEG, "record layout");
};

// one wave per (local row block, tree): grid (n_blocks, trees in batch), 64 threads.  As
// k_block_records_mono with the seeds of 4 NG expansion segments.
template <int NG>
__global__ __launch_bounds__(64) void k_block_records_wide(
    const int64_t *__restrict__ tree_off, int t0, int n_batch, const int32_t *__restrict__ pos,
    int64_t npad, const int64_t *__restrict__ st_off, const double *__restrict__ stv,
    int row_begin, int row_end, unsigned char *__restrict__ rec_all) {
    using L = wide_layout<NG>;
    const int blk = blockIdx.x;
    const int tl = blockIdx.y;
    const int lane = threadIdx.x;
    const int m = (int)(tree_off[t0 + tl + 1] - tree_off[t0 + tl]) - 1;
    const int row = row_begin + blk * SCS_TR + lane;
    int p = -1;
    if (row < row_end) p = pos[(int64_t)tl * npad + row];
    const u32 pk = p < 0 ? 0x7FFFFFFFu : (u32)p;
    u64 key = ((u64)pk << 32) | (u32)lane;
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
    double g = 0.0;
    int argpos = 0;
    if (lane < cnt - 1) {
        g = rmq_min<double>(st, m, spos, next_pos);
        int lo = spos, hi = next_pos;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rmq_min<double>(st, m, lo, mid) == g) hi = mid;
            else lo = mid;
        }
        argpos = lo;
    }
    unsigned char *rec = rec_all + ((int64_t)blk * n_batch + tl) * L::BYTES;
    ((int *)(rec + L::SPOS))[lane] = spos;
    ((double *)(rec + L::G))[lane] = g;
    ((int *)(rec + L::ARGPOS))[lane] = argpos;
    if ((lane & 7) == 7) ((int *)(rec + L::PIV))[lane >> 3] = spos;
    rec[L::SORIG + lane] = (unsigned char)orig;
    rec[L::RANK + orig] = (unsigned char)lane;
    if (lane == 0) {
        *(int *)(rec + L::CNT) = cnt;
        *(int *)(rec + L::M) = m;
        *(int64_t *)(rec + L::STOFF) = st_off[tl];
    }
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double *seed = (double *)(rec + L::SEED);
#pragma unroll
    for (int w = 1; w < L::NSEG; ++w) {
        const int end = L::seg(w);
        double mine = lane < end ? g : inf;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double other = __shfl_down(mine, off, 64);
            if (lane + off < end) mine = min_f64(mine, other);
        }
        seed[(w - 1) * 64 + lane] = mine;
    }
}

typedef int scs_int4 __attribute__((ext_vector_type(4)));

struct wide_params {
    mono_params m;
    const int4 *groups;  // x, y: indices into m.tiles of up to two tiles of ONE row block (-1: none)
    int producer_prio;   // s_setprio of the producer waves (0: as the consumers)
};

template <bool SYM>
__device__ __forceinline__ void tile_store_wide(const mono_params &p, double (&acc)[SCS_TR],
                                                const int2 tile, const bool active, const int slot,
                                                const int row0, const int col, const int self,
                                                const int ltid, const int lane, const int wl,
                                                double *t) {
    if (self >= 0) {
#pragma unroll
        for (int i = 0; i < SCS_TR; ++i)
            if (i == self) acc[i] = 0.0;
    }
    if (p.tile_out) {
        if (active) {
            double *tp = p.tile_out + (int64_t)slot * SCS_TR * MONO_TCW + ltid;
#pragma unroll
            for (int i = 0; i < SCS_TR; ++i) tp[i * MONO_TCW] = acc[i];
        }
        return;
    }
    if (active && col < p.n) {
#pragma unroll
        for (int i = 0; i < SCS_TR; ++i) {
            const int r = row0 + i;
            if (r < p.row_end) p.w[(int64_t)(r - p.row_begin) * p.ld + col] = acc[i];
        }
    }
    if (SYM && p.mirror) {
        // mirror image through LDS, eight rows at a time (tile_store of scs_mono.h); every wave of
        // the workgroup joins the barriers, a sub-tile works in its own staging region `t`
        const bool wave_mirrors =
            active && ((row0 / MONO_TCW) + 1) * MONO_TCW <= ((tile.y * MONO_TCW + wl * 64) / SCS_TR) * SCS_TR;
#pragma unroll
        for (int q = 0; q < SCS_TR / 8; ++q) {
            SCS_BARE_BARRIER();
#pragma unroll
            for (int j = 0; j < 8; ++j) t[ltid * 9 + j] = acc[q * 8 + j];
            SCS_BARE_BARRIER();
            if (wave_mirrors) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int cl = wl * 64 + it * 8 + (lane >> 3);
                    const int c = tile.y * MONO_TCW + cl;
                    const int r = row0 + q * 8 + (lane & 7);
                    if (c < p.n && r < p.row_end) p.w[(int64_t)c * p.ld + r] = t[cl * 9 + (lane & 7)];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_accumulate_spec: producer and consumer waves
// ---------------------------------------------------------------------------
// A wave that searches its column among the tile's rows, issues the range-minimum loads and
// waits for them walks a chain of dependent LDS and memory round trips during which it issues
// no cell reads -- and the waves of a wide workgroup do that in step.  Here the roles are
// split: of a workgroup's twelve waves EIGHT (two tiles of one row block) are
// consumers -- per tree: read the column's (table row address, own value) pair from LDS, then the
// one woven statement (cells of tree t + the wave's expansion steps of tree t + 1) -- and FOUR are
// producers: each runs the column step of k_accumulate_mono for the 128 columns of two consumer
// waves, THREE trees ahead (a step is about 2 us, a query that misses the L2 takes as long; three
// sets of answers in flight), and leaves the pairs in LDS; they also stage the records.  One barrier
// per tree for all twelve.  The kernel wants LONG tree batches (scs_pcg_build gives it up to 512
// trees): its queries are off the cells' critical path, so the cache locality of a short batch
// buys it little, while every launch costs a one-workgroup-per-CU kernel a prologue, tile loads /
// stores and a thin last round.  Measured (tools/acc_ab.py, one MI355X, same process and tables as
// k_accumulate_mono; profiles/r04_spec_long_batches.txt, r04_spec_batch_sweep.txt):
//   10 000 taxa /   500 trees   6.1 ms against 7.0   (4.4e12 cell-trees/s against 3.85e12)
//   50 000 taxa / 2 000 trees   600 ms against 681   (4.2e12 against 3.7e12)
//  100 000 taxa / 5 000 trees   6.52 s against 7.37  (3.85e12 against 3.4e12)
// scs_pcg_build takes it whenever its groups fill the chip (SCS_WIDE=0 / 1 force either kernel) and
// leaves tree batches shorter than 96 trees (the first 64 trees of tables still on their way) to
// the 4-wave kernel.
struct spec_layout {
    using L = wide_layout<2>;
    static constexpr int CONSUMERS = 8, PRODUCERS = 4, THREADS = 64 * (CONSUMERS + PRODUCERS);
    static constexpr size_t O_T = 0;                                      // double[2][DT_DOUBLES]
    static constexpr int NREC = 6;  // a record is in LDS from step t - 5 (stored) to step t - 1 (the consumers' expansion state)
    static constexpr size_t O_REC = 2 * (size_t)DT_DOUBLES * 8;          // [NREC][L::BYTES]
    static constexpr size_t O_VN = O_REC + NREC * (size_t)L::BYTES;      // double[2][512]
    static constexpr size_t O_ADDR = O_VN + 2 * 512 * 8;                 // unsigned[2][512]
    static constexpr size_t LDS_BYTES = O_ADDR + 2 * 512 * 4;
    static_assert(O_VN % 16 == 0 && 3 * (size_t)MONO_TCW * 9 * 8 <= LDS_BYTES, "layout");
};

template <bool SYM, bool STAMPED>
__global__ __launch_bounds__(spec_layout::THREADS) void k_accumulate_spec(wide_params wp) {
    using S = spec_layout;
    using L = wide_layout<2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
    double *const s_t = (double *)(s_mem + S::O_T);
    unsigned char *const s_rec = s_mem + S::O_REC;
    double *const s_vn = (double *)(s_mem + S::O_VN);
    unsigned *const s_addr = (unsigned *)(s_mem + S::O_ADDR);
    const mono_params &p = wp.m;

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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= S::CONSUMERS;
    const int pw = wave - S::CONSUMERS;  // producer index (0..3): the columns of consumer waves 2 pw, 2 pw + 1
    const int4 grp = wp.groups[blockIdx.x];
    const int nt = p.n_batch;
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    // consumer geometry (a producer takes that of the first of its two consumer waves)
    const int cw = producer ? 2 * pw : wave;
    const int sub = cw >> 2, wl = cw & 3;
    const int ltid = (cw & 3) * 64 + lane;
    const int ti = sub == 0 ? grp.x : grp.y;
    const bool active = ti >= 0;
    const int2 tile = p.tiles[active ? ti : grp.x];
    const int blk = tile.x;
    const int row0 = p.row_begin + blk * SCS_TR;
    const int col = tile.y * MONO_TCW + ltid;
    const int self = (col >= row0 && col < row0 + SCS_TR && col < p.row_end) ? col - row0 : -1;

    const unsigned char *rec_base = p.rec + (int64_t)blk * nt * L::BYTES;
    const __amdgpu_buffer_rsrc_t r_pos =
        __builtin_amdgcn_make_buffer_rsrc((void *)p.pos, 0, (int)(nt * p.npad * 4), 0x00020000);
    const int lane16 = lane * 16;

    // ---------------- producer side ----------------
    // record of tree t lives in s_rec[t % NREC]; producer pw copies pieces pw and pw + 4 -- THROUGH
    // REGISTERS, loaded three steps before they are stored.  (The first version sent the pieces
    // straight to LDS -- buffer_load ... lds in inline asm, invisible to the compiler, ordered by
    // hand-counted s_waitcnt vmcnt -- and the compiler, counting only the loads it could see, put
    // waits in front of the uses of the range-minimum values and the positions that were too
    // strict by the number of DMA operations in flight: a search waited for loads one step old.
    // Harmless while those hit the L2 -- 10 000 leaves --, a quarter of the step at 50 000.  With
    // every vector-memory operation visible the compiler's own counts are exact: loads return in
    // order, a use waits for exactly the loads issued before its own.)
    const __amdgpu_buffer_rsrc_t r_rec =
        __builtin_amdgcn_make_buffer_rsrc((void *)rec_base, 0, nt * L::BYTES, 0x00020000);
    struct pieces {
        scs_int4 v[2];
    };
    auto load_record = [&](int t, pieces &r) {
        if (t >= nt) return;
#pragma unroll
        for (int pc = 0; pc < L::PIECES; pc += S::PRODUCERS) {
            const int piece = pc + pw;
            if (piece < L::PIECES) {
                const int left = (L::BYTES - piece * 1024) / 16;
                if (lane < left)
                    r.v[pc / S::PRODUCERS] =
                        __builtin_amdgcn_raw_buffer_load_b128(r_rec, lane16, t * L::BYTES + piece * 1024, 0);
            }
        }
    };
    auto store_record = [&](int t, const pieces &r) {
        if (t >= nt) return;
#pragma unroll
        for (int pc = 0; pc < L::PIECES; pc += S::PRODUCERS) {
            const int piece = pc + pw;
            if (piece < L::PIECES) {
                const int left = (L::BYTES - piece * 1024) / 16;
                if (lane < left)
                    *(scs_int4 *)(s_rec + (t % S::NREC) * L::BYTES + piece * 1024 + lane16) = r.v[pc / S::PRODUCERS];
            }
        }
    };
    // the two columns of this lane (one per consumer wave served) and their positions in the
    // tree the next column step is for
    // (cpos: in the tree the next search is for; cpos2: in the tree after that -- a position is
    // requested TWO searches ahead and in front of the search's own table loads: loads return in
    // order, so a search that waits for positions requested at the end of the search before it
    // waits for that search's range-minimum loads too, and at 50 000 leaves those come from HBM)
    int pcol4[2], pself[2], cpos[2], cpos2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = tile.y * MONO_TCW + (((2 * pw + k) & 3) * 64) + lane;
        pcol4[k] = c * 4;
        pself[k] = (c >= row0 && c < row0 + SCS_TR && c < p.row_end) ? c - row0 : -1;
        cpos[k] = -1;
        cpos2[k] = -1;
    }
    // the column step of k_accumulate_mono for both columns, in two halves a whole step apart:
    // `search` (tree t) finds the neighbours and ISSUES the one range-minimum query per column --
    // four loads, in flight across the barrier -- and requests the positions in tree t + 1;
    // `finish` (TWO steps later: a step is about 2 us, a query that misses the L2 takes as long)
    // turns the answers into the pair (table row address, own value) in slot t & 1 of the
    // hand-off arrays
    struct query {
        double qx[2], qy[2];
        int cstate[2];
    };
    // the record's pivots and scalars come through the SCALAR path, from the record's copy in device
    // memory (constant for the kernel's duration: address space 4, a uniform address -> s_load): one
    // LDS round trip less in a search whose every LDS access queues behind the consumers' cell
    // reads.  They are requested a whole step before the search that needs them (`tail`, twelve
    // SGPRs carried from one search to the next): the records of a batch are gigabytes at 50 000
    // leaves, the load misses every cache, and waiting for it inside the search (1-2 us of a 2 us
    // step) made the producers the slower side there.
    typedef const __attribute__((address_space(4))) int *cint;
    int tail[12];
    auto request_tail = [&](int t) {
        cint gpiv = (cint)(size_t)(rec_base + (int64_t)min(t, nt - 1) * L::BYTES + L::PIV);
#pragma unroll
        for (int k = 0; k < 12; ++k) tail[k] = gpiv[k];
    };
    auto search = [&](int t_raw, query &q) {
        const int t = min(t_raw, nt - 1);
        const unsigned char *rb = s_rec + (t % S::NREC) * L::BYTES;
        const int *s_spos = (const int *)(rb + L::SPOS);
        const int *s_arg = (const int *)(rb + L::ARGPOS);
        const unsigned char *s_sorig = rb + L::SORIG;
        // (the searches run over consecutive trees: `tail` holds tree t's, requested by the search before)
        const int4 pa = make_int4(tail[0], tail[1], tail[2], tail[3]);
        const int4 pb = make_int4(tail[4], tail[5], tail[6], tail[7]);
        const int cnt = tail[8], m = tail[9];  // (CNT and M follow the pivots, STOFF follows them)
        const unsigned so_lo = (unsigned)tail[10], so_hi = (unsigned)tail[11];
        static_assert(L::CNT == L::PIV + 32 && L::M == L::PIV + 36 && L::STOFF == L::PIV + 40, "record tail");
        const unsigned char *st = (const unsigned char *)(p.stv + (((u64)so_hi << 32) | so_lo));
        // this search's positions; the next one's move up; the one after that is requested now
        int cp_now[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            cp_now[k] = cpos[k];
            cpos[k] = cpos2[k];
            cpos2[k] = __builtin_amdgcn_raw_buffer_load_b32(r_pos, pcol4[k], min(t + 2, nt - 1) * (int)p.npad * 4, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int cp = cp_now[k];
            const bool present = cp >= 0 && cnt > 0;
            int lo = ((pa.x < cp) + (pa.y < cp) + (pa.z < cp) + (pa.w < cp) + (pb.x < cp) + (pb.y < cp) +
                      (pb.z < cp) + (pb.w < cp)) * 8;
            const int base = min(lo, 56);
            const int4 qa = *(const int4 *)&s_spos[base];
            const int4 qb = *(const int4 *)&s_spos[base + 4];
            const int l2 = base + (qa.x < cp) + (qa.y < cp) + (qa.z < cp) + (qa.w < cp) + (qb.x < cp) +
                           (qb.y < cp) + (qb.z < cp);
            lo = lo == 64 ? 64 : l2;
            const bool hasl = present && pself[k] < 0 && lo > 0;
            const bool hasr = present && pself[k] < 0 && lo < cnt;
            const int il = max(lo - 1, 0), ir = min(lo, 63);
            // (both neighbours' entries in ONE round trip, chosen afterwards: every LDS access of a
            // producer queues behind the consumers' cell reads -- ~700 clocks a trip at 50 000 leaves,
            // and three dependent ones made the search the longest thing in the step)
            const int a_il = s_arg[il];
            const int sp_il = s_spos[il], sp_ir = s_spos[ir];
            const int so_il = s_sorig[il], so_ir = s_sorig[ir];
            const bool left = hasl && (!hasr || a_il >= cp);
            const int q_anchor = left ? sp_il : sp_ir;
            const int nbrow = left ? so_il : so_ir;
            q.cstate[k] = nbrow | ((hasl || hasr) ? 256 : 0) | ((present && pself[k] >= 0) ? 512 : 0);
            const bool any = hasl || hasr;
            int o[2];
            rmq_offsets(m, any ? (left ? q_anchor : cp) : 0, any ? (left ? cp : q_anchor) : 1, o);
            q.qx[k] = *(const double *)(st + (unsigned)o[0] * 8u);
            q.qy[k] = *(const double *)(st + (unsigned)o[1] * 8u);
        }
        request_tail(t_raw + 1);
    };
    auto finish = [&](int t, const query &q) {
        const unsigned tbase =
            (unsigned)(size_t)(__attribute__((address_space(3))) double *)&s_t[(t & 1) * DT_DOUBLES];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int nb = 0;
            double vn = 0.0;
            if (q.cstate[k] & 256) {
                vn = min_f64(q.qx[k], q.qy[k]);
                nb = q.cstate[k] & 255;
            } else if (q.cstate[k] & 512) {
                nb = pself[k];
                vn = inf;
            }
            const int slot = (t & 1) * 512 + (2 * pw + k) * 64 + lane;
            s_vn[slot] = vn;
            s_addr[slot] = tbase + (unsigned)nb * (DV_LD * 8);
        }
    };

    // ---------------- consumer side ----------------
    auto expand_whole = [&](int t) {
        const unsigned char *rb = s_rec + (t % S::NREC) * L::BYTES;
        double *dv = s_t + (t & 1) * DT_DOUBLES;
        const int b0 = L::seg(wave);
        const double g_rank = ((const double *)(rb + L::G))[lane];
        const int so_rank = rb[L::SORIG + lane];
        const int rho = rb[L::RANK + lane];
        double cur = inf;
        if (rho < b0) cur = ((const double *)(rb + L::SEED))[(wave - 1) * 64 + rho];
        double *row_a = &dv[lane * DV_LD];
        double *col_a = &dv[lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = b0 + j;
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

    // Two code paths from here on, one per role, with the same sequence of barriers.  The
    // accumulators exist in the consumers' path only: were they in scope of the producers' loop,
    // 128 registers would be held through it, the queries would spill, and every reload of a spill
    // is a vector-memory operation behind an s_waitcnt vmcnt(0) -- which drains the very loads the
    // producer is there to keep in flight.  (And ONE cell statement in the kernel: a second one with
    // 64 tied accumulators makes the register allocator keep two sets.)
    if (producer) {
        // ---- prologue: records 0 ... 4 into LDS, 5 ... 7 on their way (one per register set), tree 0's
        // column pairs, the queries of trees 1, 2 and 3
        pieces ra = {{{0, 0, 0, 0}, {0, 0, 0, 0}}}, rb = ra, rc = ra;
        {
            pieces first[5] = {ra, ra, ra, ra, ra};  // (all five loads in flight together)
#pragma unroll
            for (int t = 0; t < 5; ++t) load_record(t, first[t]);
#pragma unroll
            for (int t = 0; t < 5; ++t) store_record(t, first[t]);
        }
        load_record(5, ra);
        load_record(6, rb);
        load_record(7, rc);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            cpos[k] = __builtin_amdgcn_raw_buffer_load_b32(r_pos, pcol4[k], 0, 0);
            cpos2[k] = __builtin_amdgcn_raw_buffer_load_b32(r_pos, pcol4[k], min(1, nt - 1) * (int)p.npad * 4, 0);
        }
        SCS_BARE_BARRIER();  // (records 0 ... 4 are stores of this wave: drained by the barrier's lgkmcnt wait)
        query qa = {{0.0, 0.0}, {0.0, 0.0}, {0, 0}}, qb = qa, qc = qa;
        request_tail(0);
        search(0, qa);
        finish(0, qa);  // (waits for tree 0's answers: once per launch)
        search(1, qa);
        search(2, qb);
        search(3, qc);
        SCS_BARE_BARRIER();
        if (STAMPED) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
        // (few instructions, all of them on the critical path of the step -- unless the consumers are)
        if (wp.producer_prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (wp.producer_prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (wp.producer_prio == 3) __builtin_amdgcn_s_setprio(3);
        // Step tl.  On entry: the queries of trees tl + 1 (in `q1`, issued three steps ago), tl + 2 and
        // tl + 3 are in flight; the three sets take turns.  (Four sets, four steps ahead: measured, no
        // faster at 50 000 leaves -- 579 against 573 ms: what the producers wait for there is not the
        // age of a load.)
        auto step = [&](int tl, query &q1, pieces &r1) __attribute__((always_inline)) {
            if (tl + 1 < nt) finish(tl + 1, q1);
            stamp(0);
            // the record of tree tl + 5 (searched in the next step), loaded three steps ago, goes to
            // LDS; its registers take the record of tree tl + 8
            store_record(tl + 5, r1);
            load_record(tl + 8, r1);
            search(tl + 4, q1);  // six loads (two positions, four table entries), consumed two / three steps on
            stamp(1);
            SCS_BARE_BARRIER();
            stamp(4);
        };
        for (int tl = 0; tl < nt; tl += 3) {
            step(tl, qa, ra);
            if (tl + 1 < nt) step(tl + 1, qb, rb);
            if (tl + 2 < nt) step(tl + 2, qc, rc);
        }
        if (STAMPED && lane == 0 && p.stamps) {
#pragma unroll
            for (int k = 0; k < 7; ++k) atomicAdd(&p.stamps[k], ts[k]);
            atomicAdd(&p.stamps[7], 1ull);
        }
        // the barriers of the consumers' mirror image
        if (!p.tile_out && SYM && p.mirror) {
#pragma unroll
            for (int q = 0; q < SCS_TR / 8; ++q) {
                SCS_BARE_BARRIER();
                SCS_BARE_BARRIER();
            }
        }
    } else {
        double acc[SCS_TR];
#pragma unroll
        for (int i = 0; i < SCS_TR; ++i) {
            double v = 0.0;
            if (active && p.load_w && p.tile_out)
                v = p.tile_out[((int64_t)ti * SCS_TR + i) * MONO_TCW + ltid];
            else if (active && p.load_w && col < p.n && row0 + i < p.row_end)
                v = p.w[(int64_t)(row0 - p.row_begin + i) * p.ld + col];
            acc[i] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SCS_BARE_BARRIER();  // the records are in place
        expand_whole(0);
        SCS_BARE_BARRIER();
        if (STAMPED) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
        // Step tl.  On entry (all waves past the barrier): the table of tree tl is complete in
        // s_t[tl & 1], its column pairs in slot tl & 1; the record of tree tl + 1 is in place.
        for (int tl = 0; tl < nt; ++tl) {
            const int slot = (tl & 1) * 512 + wave * 64 + lane;
            const double vn = s_vn[slot];
            const unsigned addr = s_addr[slot];
            const unsigned char *rb = s_rec + (min(tl + 1, nt - 1) % S::NREC) * L::BYTES;
            const double g_rank = ((const double *)(rb + L::G))[lane];
            const int so_rank = rb[L::SORIG + lane];
            const int rho = rb[L::RANK + lane];
            const int b0 = L::seg(wave);
            double cur = inf;
            if (rho < b0) cur = ((const double *)(rb + L::SEED))[(wave - 1) * 64 + rho];
            double *dv = s_t + ((tl + 1) & 1) * DT_DOUBLES;
            const unsigned rowb = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&dv[lane * DV_LD];
            const unsigned colb = (unsigned)(size_t)(__attribute__((address_space(3))) double *)&dv[lane];
            stamp(2);
            double tmp[SCS_CELLS_DEPTH];
            unsigned x1, x2;
            SCS_CELLS_EXPAND_ASM(acc, tmp, x1, x2, addr, vn, cur, (int)__double2loint(g_rank),
                                 __double2hiint(g_rank), so_rank, rho, rowb, colb, b0, DV_LD * 8);
            stamp(3);
            SCS_BARE_BARRIER();
            stamp(4);
        }
        if (STAMPED && lane == 0 && p.stamps) {
#pragma unroll
            for (int k = 0; k < 7; ++k) atomicAdd(&p.stamps[k], ts[k]);
            atomicAdd(&p.stamps[7], 1ull);
        }
        tile_store_wide<SYM>(p, acc, tile, active, ti, row0, col, self, ltid, lane, wl,
                             (double *)s_mem + (size_t)sub * MONO_TCW * 9);
    }
}
