// Matrix-free operator: Y = W Z straight from the flattened tables, W never formed.
//
// A MEASURED COMPARISON, not the product path (round 5, the review's item 8): north_star prescribes the
// dense W, scs_pcg_build + the SYMM stream stay the headline.  What this variant shows is the price of an
// operator application when the O(V^2 M) build is skipped altogether, and how far its Fiedler pair lies from
// the dense path's (tools/matrix_free_compare.py, profiles/r05_matrix_free_comparison.json).
//
// Reference semantics (src/sc_supertree/scs.py:569-663 as restated in oracle/pcg_oracle.c): a pair of
// leaves (a, b) of tree t whose lowest common ancestor is not the root adds value(LCA) * weight(t) to
// W[a][b]; the LCA of the leaves at DFS positions p < q is the separator of minimum depth in
// adj_depth[p .. q-1], its value the adj_val there.  Hence, per tree,
//     (W_t z)_p = sum_{q != p} v(p, q) z_q ,   v(p, q) = value at the shallowest separator between p and q,
// and the sum over q < p is a classic running quantity: walking the leaves left to right, the leaves seen so
// far fall into groups by the depth of their LCA with the current leaf -- one group per ancestor on the
// current root path that has leaves to the left -- and
//     L_p = sum over groups  value(group) * (sum of z over the group).
// A separator of depth d between p and p + 1 merges every group of depth >= d (and leaf p) into the group of
// depth d: a stack of (depth, value, sum), each leaf pushed and popped once.  The sum over q > p is the same
// walk from the right.  One thread per (tree, direction, column): the stack's top lives in registers, the
// rest in LDS and, beyond MF_LDS_LEVELS entries, in a per-thread strip of global memory (its depth is
// bounded by the tree's deepest separator).  Every
// thread scatters its results to its own slab Y[direction][tree][taxon][column]; k_mf_reduce adds the slabs
// of a taxon in tree order (left then right per tree): deterministic, no atomics.
//
// Rounding differs from the dense path (there: W[a][b] is the rounded sum over trees of rounded products,
// then W z; here: per tree a running total updated by fused multiply-adds, then the sum over trees), so the
// two agree to rounding, not bit for bit -- one more reason this is a comparison and not the product.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct mf_params {
    const int64_t *tree_off;
    const int32_t *leaf_taxon;
    const int32_t *adj_depth;
    const double *adj_val;
    const double *tree_w;
    const int64_t *stack_off;  // [n_trees + 1] entries of a tree's strip
    int32_t n_trees;
    int32_t n_taxa;
    int64_t stack_total;       // entries of all strips of ONE (direction, column, chunk)
    double *st_val;            // [2 * B * chunks * stack_total]
    double *st_sum;
    int32_t *st_dep;
    const double *x;           // operand, row-major [n_taxa][B]
    double *y_slabs;           // [2][n_trees][n_taxa][B]
    // the chunked walk (below): strips and per-(unit, tree) slots of the summaries and the carried stacks
    int32_t chunks;
    int32_t *sm_dep, *cy_dep;
    double *sm_val, *sm_sum, *cy_pa, *cy_ps;
    int32_t *sm_cnt, *sm_root, *cy_cnt;
};

// deepest separator of every tree (sizes the strips): a thread takes 64 consecutive leaf slots and sends one
// atomic per tree it meets (one atomic per slot: 155 ms at configs[3], all on 2 000 addresses)
__global__ void k_mf_maxdepth(const int64_t *__restrict__ tree_off, int n_trees,
                              const int32_t *__restrict__ adj_depth, int64_t n_leaves, int32_t *maxd) {
    const int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 64;
    if (p0 >= n_leaves) return;
    int lo = 0, hi = n_trees;  // the tree of slot p0: last t with tree_off[t] <= p0
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tree_off[mid] <= p0) lo = mid;
        else hi = mid;
    }
    int64_t next = tree_off[lo + 1];
    int best = 0;
    const int64_t p1 = p0 + 64 < n_leaves ? p0 + 64 : n_leaves;
    for (int64_t p = p0; p < p1; ++p) {
        while (p >= next) {  // into the next tree (trees without leaves: skipped)
            if (best > 0) atomicMax(&maxd[lo], best);
            best = 0;
            ++lo;
            next = tree_off[lo + 1];
        }
        if (p + 1 < next) best = max(best, adj_depth[p]);  // a tree's last slot holds no separator
    }
    if (best > 0) atomicMax(&maxd[lo], best);
}

// ---- the sweep, cut into chunks
// M * 2 * B threads walking 10^4 ... 10^5 leaves each leave the chip nearly empty, and every walk is one
// dependent chain (measured: 2 100 clocks a leaf with the stack in LDS and the leaf arrays read eight ahead,
// 4 500 with the stack in global memory; profiles/r05_matrix_free_comparison_v1/_v2.json).  So a tree's leaves
// are cut into `chunks` pieces per direction and the walk is done in three steps:
//   A  k_mf_chunk<B, false>: every chunk walks its own leaves from an EMPTY stack (no output) through the
//      separator behind its last leaf; the stack it ends with -- the chunk's leaves grouped by the depth of
//      their LCA with the next chunk's first leaf -- is its SUMMARY (plus: did it meet a root separator).
//   B  k_mf_carry<B>: one thread per (tree, direction, column) folds the summaries in order: the stack S0 a
//      chunk would have STARTED with had the walk been sequential.  Entries of S0 at least as deep as a
//      summary's shallowest entry join that entry; a root separator drops S0.  S0 is stored by prefix sums
//      over its entries in order of depth: PA = sum of value * sum, PS = sum of sums.
//   C  k_mf_chunk<B, true>: every chunk walks again from an empty stack, and a leaf's result is its local
//      total plus what S0 contributes: with m the shallowest separator between the chunk's first leaf and this
//      one (a running minimum, so the position in S0 only moves down) the groups of S0 no deeper than m keep
//      their own value -- PA up to there -- and the deeper ones have joined the node of that separator:
//      (PS_all - PS up to there) * value(m).
// Twice the walking, `chunks` times the threads.  One wave per workgroup; a thread's stack lives in LDS
// ([level][lane]: the lanes of a wave hit different banks) up to MF_LDS_LEVELS entries, deeper ones in its
// strip of global memory (a caterpillar's depth is its leaf count).  The leaf arrays are read MF_CHUNK leaves
// ahead -- taxon, separator and the operand row behind the taxon are independent of the stack, only the
// stack walk is a dependent chain.
// The B lanes of a (tree, direction, chunk) group walk in lockstep and share depth and value of every stack
// entry: one copy per group, written by the group's first lane.  22 KB of LDS a wave:
// seven waves a CU instead of three (the walk is latency-bound: more waves is what it needs).
constexpr int MF_LDS_LEVELS = 32;
constexpr int MF_CHUNK = 4;

template <int B, bool OUT>
__global__ __launch_bounds__(64, 2) void k_mf_chunk(mf_params a) {
    constexpr int G = 64 / B;
    __shared__ double l_val[MF_LDS_LEVELS][G];
    __shared__ double l_sum[MF_LDS_LEVELS][64];
    __shared__ int l_dep[MF_LDS_LEVELS][G];
    const int lane = threadIdx.x;
    const int grp = lane / B;
    const int C = a.chunks;
    const int64_t gid = (int64_t)blockIdx.x * 64 + lane;
    const int k = (int)(gid % B);
    const int dir = (int)((gid / B) & 1);
    const int c = (int)((gid / (2 * B)) % C);
    const int64_t t64 = gid / ((int64_t)2 * B * C);
    if (t64 >= a.n_trees) return;
    const int t = (int)t64;
    const int64_t off = a.tree_off[t];
    const int L = (int)(a.tree_off[t + 1] - off);
    const int CL = (L + C - 1) / C;
    const int i_begin = c * CL, i_end = min(L, i_begin + CL);
    const int64_t unit = (int64_t)(dir * B + k) * C + c;
    const int64_t strip = unit * a.stack_total + a.stack_off[t];
    const int64_t slot = unit * a.n_trees + t;
    if (i_begin >= L) {  // a tree with fewer leaves than chunks
        if (!OUT) {
            a.sm_cnt[slot] = 0;
            a.sm_root[slot] = 0;
        }
        return;
    }
    const double wt = a.tree_w[t];
    const int32_t *tax = a.leaf_taxon + off;
    const int32_t *dep = a.adj_depth + off;
    const double *val = a.adj_val + off;
    double *sv = a.st_val + strip;
    double *ss = a.st_sum + strip;
    int32_t *sd = a.st_dep + strip;
    double *out = a.y_slabs + ((int64_t)dir * a.n_trees + t) * (int64_t)a.n_taxa * B + k;
    const double *x = a.x + k;

    // what the chunks in front contribute (step C)
    const int32_t *cdep = a.cy_dep + strip;
    const double *cpa = a.cy_pa + strip;
    const double *cps = a.cy_ps + strip;
    int jp = OUT ? a.cy_cnt[slot] : 0;  // entries of S0 no deeper than the running minimum m
    const double ps_all = (OUT && jp > 0) ? cps[jp - 1] : 0.0;
    double pa_cur = (OUT && jp > 0) ? cpa[jp - 1] : 0.0, ps_cur = ps_all;
    int m = 0x7FFFFFFF;
    double vmin = 0.0;

    double total = 0.0;           // sum over the stack of value * sum
    int td = -1;                  // the top entry, in registers (td < 0: empty stack)
    double tv = 0.0, ts = 0.0;
    int sp = 0;                   // entries below the top: LDS levels [0, MF_LDS_LEVELS), then the strip
    int root_seen = 0;
    // the separators this walk processes: step A goes through the one behind the chunk's last leaf
    const int sep_end = OUT ? i_end - 1 : min(i_end, L - 1);
    // three batches in flight: the leaf arrays two batches ahead, the operand rows (behind the taxa) one
    // batch ahead -- a walk that waited for every batch's loads spent most of its time on HBM latency
    // (16 streams a wave, a new line in one of them at nearly every batch)
    int n_tax[MF_CHUNK], n_dep[MF_CHUNK], c_tax[MF_CHUNK], c_dep[MF_CHUNK];
    double n_val[MF_CHUNK], c_val[MF_CHUNK], c_x[MF_CHUNK];
    auto load_leaves = [&](int i0, int (&ltax)[MF_CHUNK], int (&ldep)[MF_CHUNK], double (&lval)[MF_CHUNK]) {
#pragma unroll
        for (int j = 0; j < MF_CHUNK; ++j) {
            const int i = i0 + j;
            const int p = dir ? L - 1 - i : i;
            const int q = dir ? p - 1 : p;
            ltax[j] = i < i_end ? tax[p] : 0;
            ldep[j] = i < sep_end ? dep[q] : 0;
            lval[j] = i < sep_end ? val[q] : 0.0;
        }
    };
    load_leaves(i_begin, c_tax, c_dep, c_val);
    load_leaves(i_begin + MF_CHUNK, n_tax, n_dep, n_val);
#pragma unroll
    for (int j = 0; j < MF_CHUNK; ++j) c_x[j] = (i_begin + j < i_end) ? x[(int64_t)c_tax[j] * B] : 0.0;
    for (int i0 = i_begin; i0 < i_end; i0 += MF_CHUNK) {
        int f_tax[MF_CHUNK], f_dep[MF_CHUNK];
        double f_val[MF_CHUNK], n_x[MF_CHUNK];
        load_leaves(i0 + 2 * MF_CHUNK, f_tax, f_dep, f_val);
#pragma unroll
        for (int j = 0; j < MF_CHUNK; ++j) n_x[j] = (i0 + MF_CHUNK + j < i_end) ? x[(int64_t)n_tax[j] * B] : 0.0;
#pragma unroll
        for (int j = 0; j < MF_CHUNK; ++j) {
            const int i = i0 + j;
            if (i < i_end) {
                if (OUT) out[(int64_t)c_tax[j] * B] = total + fma(vmin, ps_all - ps_cur, pa_cur);
                if (i < sep_end) {
                    const int d = c_dep[j];
                    const double v = d > 0 ? c_val[j] * wt : 0.0;  // one rounded multiply (scs.py:656)
                    if (OUT && d < m) {
                        m = d;
                        vmin = v;
                        while (jp > 0 && cdep[jp - 1] > m) --jp;
                        pa_cur = jp > 0 ? cpa[jp - 1] : 0.0;
                        ps_cur = jp > 0 ? cps[jp - 1] : 0.0;
                        asm volatile("" ::"v"(pa_cur), "v"(ps_cur));  // (as below: the wait stays in here)
                    }
                    double s = c_x[j];
                    while (td >= d) {  // groups of depth >= d join the group of depth d
                        s += ts;
                        total = fma(-tv, ts, total);
                        if (sp > 0) {
                            --sp;
                            if (sp < MF_LDS_LEVELS) {
                                td = l_dep[sp][grp];
                                tv = l_val[sp][grp];
                                ts = l_sum[sp][lane];
                            } else {
                                td = sd[sp - MF_LDS_LEVELS];
                                tv = sv[sp - MF_LDS_LEVELS];
                                ts = ss[sp - MF_LDS_LEVELS];
                                // (use the values HERE: the wait for these loads then sits inside this rare
                                // branch; left to the merge point it is a vmcnt(0) in every turn of the
                                // loop, i.e. a wait for the batches in flight -- measured: 6 400 clocks a leaf)
                                asm volatile("" ::"v"(td), "v"(tv), "v"(ts));
                            }
                        } else {
                            td = -1;
                        }
                    }
                    if (d > 0) {
                        if (td >= 0) {
                            if (sp < MF_LDS_LEVELS) {
                                if (lane % B == 0) {
                                    l_dep[sp][grp] = td;
                                    l_val[sp][grp] = tv;
                                }
                                l_sum[sp][lane] = ts;
                            } else {
                                sd[sp - MF_LDS_LEVELS] = td;
                                sv[sp - MF_LDS_LEVELS] = tv;
                                ss[sp - MF_LDS_LEVELS] = ts;
                            }
                            ++sp;
                        }
                        td = d;
                        tv = v;
                        ts = s;
                        total = fma(tv, ts, total);
                    } else {
                        total = 0.0;  // the root separates everything seen so far from everything to come
                        root_seen = 1;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MF_CHUNK; ++j) {
            c_tax[j] = n_tax[j];
            c_dep[j] = n_dep[j];
            c_val[j] = n_val[j];
            c_x[j] = n_x[j];
            n_tax[j] = f_tax[j];
            n_dep[j] = f_dep[j];
            n_val[j] = f_val[j];
        }
    }
    if (!OUT) {
        // the summary, shallowest entry first
        int32_t *md = a.sm_dep + strip;
        double *mv = a.sm_val + strip, *ms = a.sm_sum + strip;
        for (int j = 0; j < sp; ++j) {
            if (j < MF_LDS_LEVELS) {
                md[j] = l_dep[j][grp];
                mv[j] = l_val[j][grp];
                ms[j] = l_sum[j][lane];
            } else {
                md[j] = sd[j - MF_LDS_LEVELS];
                mv[j] = sv[j - MF_LDS_LEVELS];
                ms[j] = ss[j - MF_LDS_LEVELS];
            }
        }
        int cnt = sp;
        if (td >= 0) {
            md[cnt] = td;
            mv[cnt] = tv;
            ms[cnt] = ts;
            ++cnt;
        }
        a.sm_cnt[slot] = cnt;
        a.sm_root[slot] = root_seen;
    }
}

// step B: the stack every chunk would have started with, by prefix sums (one thread per tree, direction, column)
template <int B>
__global__ __launch_bounds__(64) void k_mf_carry(mf_params a) {
    const int C = a.chunks;
    const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int k = (int)(gid % B);
    const int dir = (int)((gid / B) & 1);
    const int64_t t64 = gid / (2 * B);
    if (t64 >= a.n_trees) return;
    const int t = (int)t64;
    // the running stack lives in the (dead between steps A and C) overflow strip of chunk 0
    const int64_t unit0 = (int64_t)(dir * B + k) * C;
    const int64_t strip0 = unit0 * a.stack_total + a.stack_off[t];
    int32_t *rd = a.st_dep + strip0;
    double *rv = a.st_val + strip0, *rs = a.st_sum + strip0;
    int cnt = 0;
    for (int c = 0; c < C; ++c) {
        const int64_t unit = unit0 + c;
        const int64_t strip = unit * a.stack_total + a.stack_off[t];
        const int64_t slot = unit * a.n_trees + t;
        int32_t *cd = a.cy_dep + strip;
        double *pa = a.cy_pa + strip, *ps = a.cy_ps + strip;
        double run_a = 0.0, run_s = 0.0;
        for (int j = 0; j < cnt; ++j) {
            run_a = fma(rv[j], rs[j], run_a);
            run_s += rs[j];
            cd[j] = rd[j];
            pa[j] = run_a;
            ps[j] = run_s;
        }
        a.cy_cnt[slot] = cnt;
        const int scnt = a.sm_cnt[slot];
        if (a.sm_root[slot]) cnt = 0;
        if (scnt > 0) {
            const int32_t *md = a.sm_dep + strip;
            const double *mv = a.sm_val + strip, *ms = a.sm_sum + strip;
            const int m = md[0];
            double extra = 0.0;
            while (cnt > 0 && rd[cnt - 1] >= m) {
                --cnt;
                extra += rs[cnt];
            }
            for (int j = 0; j < scnt; ++j) {
                rd[cnt] = md[j];
                rv[cnt] = mv[j];
                rs[cnt] = ms[j] + (j == 0 ? extra : 0.0);
                ++cnt;
            }
        }
    }
}

// y[taxon][k] = sum over trees, in order, of (left slab + right slab)
template <int B>
__global__ __launch_bounds__(256) void k_mf_reduce(const double *__restrict__ slabs, int n_trees, int n_taxa,
                                                   double *__restrict__ y) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)n_taxa * B;
    if (e >= per) return;
    const double *l = slabs + e;
    const double *r = slabs + (int64_t)n_trees * per + e;
    double acc = 0.0;
    int t = 0;
    for (; t + 4 <= n_trees; t += 4) {  // four trees' loads in flight, added in order
        const double l0 = l[(int64_t)t * per], r0 = r[(int64_t)t * per];
        const double l1 = l[(int64_t)(t + 1) * per], r1 = r[(int64_t)(t + 1) * per];
        const double l2 = l[(int64_t)(t + 2) * per], r2 = r[(int64_t)(t + 2) * per];
        const double l3 = l[(int64_t)(t + 3) * per], r3 = r[(int64_t)(t + 3) * per];
        acc += l0 + r0;
        acc += l1 + r1;
        acc += l2 + r2;
        acc += l3 + r3;
    }
    for (; t < n_trees; ++t) acc += l[(int64_t)t * per] + r[(int64_t)t * per];
    y[e] = acc;
}

// k-major operand (b x ldz) -> row-major [n][B]
template <int B>
__global__ void k_mf_operand(const double *__restrict__ zt, int64_t ldz, int n, double *__restrict__ x) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * B) return;
    const int r = (int)(e / B), k = (int)(e % B);
    x[e] = zt[(int64_t)k * ldz + r];
}

__global__ void k_mf_fill(double *p, int64_t n, double v) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) p[e] = v;
}

__global__ void k_mf_column0(const double *__restrict__ y, int n, int b, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = y[(int64_t)i * b];
}
