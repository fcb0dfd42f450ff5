// Source forests resident in HBM, and their restriction on the device (round 5).
//
// The recursion of construct_supertree restricts every source tree to every part of a node's
// partition (reference: src/sc_supertree/scs.py:139-171 with :411-455, cogent3's get_sub_tree per
// tree and part).  Rounds 2-4 did that on the host (csrc/scs_host.c: one preorder sweep per tree
// serves all parts; then scs_host_flatten turns each child forest into the tables of
// include/scs_hip.h) -- 12 ns per tree node and level, which at configs[4]'s shape (10^9 tree
// nodes per level, 61 000 nodes of 5 000 trees each) was three quarters of the whole recursion.
// Here the forest stays in device memory as the same preorder node arrays and a split is three
// launches:
//   k_split_count   one THREAD per tree: subtree ends, the sweep with the root path on a stack
//                   (the LCA of a part's consecutive leaves = the deepest stack entry at or before
//                   the earlier leaf: binary search), one mark bit per part on every node a part
//                   keeps; leaves and nodes per (part, tree);
//   k_split_scan    one workgroup: exclusive scans over the trees -- child tree index, node and leaf
//                   offsets, totals, per part;
//   k_split_fill    one thread per tree again: per part the marked nodes in index order = the
//                   restricted tree in preorder; parents by the interval test, merged branch
//                   lengths folded bottom-up with the parent's length in front (the SAME order of
//                   additions as scs_host.c / tree.py:get_sub_tree: same bits), taxa renumbered --
//                   and, in the same pass, the child's TABLES (leaf_taxon, adj_depth, adj_val by
//                   weighting strategy, as scs_host_flatten: same bits), weights, present taxa.
// Trees are independent, so the parallelism is across trees (thousands); a thread walks its tree
// alone.  The host keeps what north_star leaves to it: labels, tie-breaks, contraction groups,
// components, tree assembly -- it downloads a child's tables (16 bytes per leaf) for that and hands
// them on exactly as the host path does.
#include <cmath>
#include <memory>

#include "scs_internal.h"

namespace {

constexpr int SPLIT_MAX_PARTS = 8;
constexpr int SPLIT_THREADS = 64;   // trees per workgroup (one thread each)
// A workgroup whose 64 trees hold at most this many nodes works on a copy of them in LDS (the
// deep recursion: thousands of trees of a dozen nodes -- a thread's sweep is a chain of dependent
// loads, 100 cycles each from LDS against 2 000 from memory: 446 -> ~100 us per split of 5 000
// trees of 47 nodes); larger trees are walked in place.
constexpr int SPLIT_CAP = 2304;
constexpr size_t SPLIT_FILL_LDS = (size_t)SPLIT_CAP * (4 + 4 + 8 + 8 + 4 + 4 + 4 + 4 + 8 + 1) + 64;

struct split_params {
    int32_t n_trees, n_parts, strategy;
    int32_t tpb;  // trees per workgroup (one thread each; 64 down to 8, so that their nodes fit SPLIT_CAP)
    const int64_t *node_off;
    const int32_t *parent, *taxon;
    const double *length, *support, *weights;
    const int32_t *part_of, *new_id;
    // scratch, sized like the parent's node arrays (used by the workgroups that walk their trees in place)
    int32_t *stk_a, *stk_b, *stk_c, *cdepth;
    unsigned char *mark;
    double *cval;
    // per (part, tree)
    int32_t *leaves_cnt, *nodes_cnt, *tree_pos;
    int64_t *node_start, *leaf_start;  // absolute positions in the children's shared arrays
    int64_t *totals;                   // [n_parts][4]: trees, nodes, leaves, first node of the part
    // children (shared arrays; a part's slice starts at totals[part][3] / leaf base)
    int64_t *c_node_off, *c_tree_off;  // [n_parts][n_trees + 1]
    int32_t *c_parent, *c_taxon, *c_tree_index;
    double *c_length, *c_support, *c_weights;
    int32_t *c_leaf_taxon, *c_adj_depth;
    double *c_adj_val;
    unsigned char *c_present;  // [n_parts][n_taxa_parent] (only the first part_taxa entries are used)
    int32_t present_ld;
    int32_t *flags;  // [0]: error code, [1 + part]: monotone (1 until a negative length is met)
};

__global__ __launch_bounds__(SPLIT_THREADS) void k_split_count(split_params p) {
    __shared__ int32_t s_last[SPLIT_MAX_PARTS][SPLIT_THREADS];
    __shared__ int32_t s_cnt[SPLIT_MAX_PARTS][SPLIT_THREADS];
    __shared__ int32_t s_nodes[SPLIT_MAX_PARTS][SPLIT_THREADS];
    __shared__ int32_t l_par[SPLIT_CAP], l_tax[SPLIT_CAP], l_st[SPLIT_CAP];
    __shared__ unsigned char l_mk[SPLIT_CAP];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * p.tpb;
    const int t1 = min(p.n_trees, t0 + p.tpb);
    const int t = t0 + tid;
    const int np = p.n_parts;
    const int64_t n0 = p.node_off[t0], n1 = p.node_off[t1];
    const bool stage = n1 - n0 <= SPLIT_CAP;
    if (stage) {
        for (int i = tid; i < (int)(n1 - n0); i += SPLIT_THREADS) {
            l_par[i] = p.parent[n0 + i];
            l_tax[i] = p.taxon[n0 + i];
            l_mk[i] = 0;
        }
        __syncthreads();
    }
    if (t < t1) {
        const int64_t off = p.node_off[t];
        const int32_t k = (int32_t)(p.node_off[t + 1] - off);
        const int32_t rel = (int32_t)(off - n0);
        const int32_t *par = stage ? l_par + rel : p.parent + off;
        const int32_t *tax = stage ? l_tax + rel : p.taxon + off;
        int32_t *st = stage ? l_st + rel : p.stk_a + off;
        unsigned char *mk = stage ? l_mk + rel : p.mark + off;
        for (int b = 0; b < np; ++b) {
            s_last[b][tid] = 0;
            s_cnt[b][tid] = 0;
            s_nodes[b][tid] = 0;
        }
        int32_t sp = 0;
        bool bad = false;
        for (int32_t i = 0; i < k; ++i) {
            const int32_t q = par[i];
            if (i > 0 && (q < 0 || q >= i)) {  // not preorder
                bad = true;
                break;
            }
            // the root path of i: the inner nodes still on the stack up to its parent
            while (sp > 0 && st[sp - 1] != q) --sp;
            const int32_t x = tax[i];
            if (x < 0) {
                st[sp++] = i;
                continue;
            }
            const int32_t pc = p.part_of[x];
            if (pc < 0) continue;
            const unsigned char bit = (unsigned char)(1u << pc);
            const int32_t xl = s_last[pc][tid] - 1;
            if (xl >= 0 && sp > 0) {
                // deepest ancestor of i whose index is <= xl: the LCA of xl and i (st[0] = root <= xl)
                int32_t lo = 0, hi = sp - 1;
                while (lo < hi) {
                    const int32_t mid = (lo + hi + 1) >> 1;
                    if (st[mid] <= xl) lo = mid;
                    else hi = mid - 1;
                }
                const int32_t a = st[lo];
                const unsigned char old = mk[a];
                if (!(old & bit)) {
                    mk[a] = old | bit;
                    s_nodes[pc][tid] += 1;
                }
            }
            mk[i] |= bit;
            s_nodes[pc][tid] += 1;
            s_last[pc][tid] = i + 1;
            s_cnt[pc][tid] += 1;
        }
        if (bad) atomicExch(&p.flags[0], SCS_EINVAL);
        // (a part with fewer than two leaves of this tree is dropped here: nothing counted)
        for (int b = 0; b < np; ++b) {
            const int32_t c = s_cnt[b][tid];
            const bool keep = c >= 2 && !bad;
            p.leaves_cnt[(int64_t)b * p.n_trees + t] = keep ? c : 0;
            p.nodes_cnt[(int64_t)b * p.n_trees + t] = keep ? s_nodes[b][tid] : 0;
        }
    }
    if (stage) {
        __syncthreads();
        for (int i = tid; i < (int)(n1 - n0); i += SPLIT_THREADS) p.mark[n0 + i] = l_mk[i];
    }
}

// one workgroup of 1024: per part, exclusive scans over the trees
__global__ __launch_bounds__(1024) void k_split_scan(split_params p) {
    // (round 5, late: wave-level scans -- shuffles inside the sixteen waves, their totals scanned by the
    // first wave, two barriers in all -- instead of a ten-step Hillis-Steele scan per part with two barriers a
    // step: 28 -> 9 us a call, and a recursion makes tens of thousands of them)
    __shared__ int64_t s_tot[SPLIT_MAX_PARTS][16][3];   // per part, per wave: trees, nodes, leaves
    __shared__ int64_t s_pre[SPLIT_MAX_PARTS][16][3];   // exclusive over the waves
    __shared__ int64_t s_base[SPLIT_MAX_PARTS + 1][2];  // node / leaf base of a part (sum over the parts before)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.n_trees;
    const int per = (M + 1023) / 1024;
    const int t0 = min(M, tid * per), t1 = min(M, t0 + per);
    int64_t own[SPLIT_MAX_PARTS][3], inc[SPLIT_MAX_PARTS][3];
#pragma unroll
    for (int b = 0; b < SPLIT_MAX_PARTS; ++b) {
        if (b >= p.n_parts) break;
        const int32_t *lc = p.leaves_cnt + (int64_t)b * M, *nc = p.nodes_cnt + (int64_t)b * M;
        int64_t a = 0, n = 0, l = 0;
        for (int t = t0; t < t1; ++t) {
            a += lc[t] > 0;
            n += nc[t];
            l += lc[t];
        }
        own[b][0] = a;
        own[b][1] = n;
        own[b][2] = l;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            int64_t x = own[b][v];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int64_t y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            inc[b][v] = x;
            if (lane == 63) s_tot[b][wave][v] = x;
        }
    }
    __syncthreads();
    if (wave == 0) {
        // lanes 0 .. 15 hold a wave's totals: exclusive scan over the waves, per part and value
        for (int b = 0; b < p.n_parts; ++b)
            for (int v = 0; v < 3; ++v) {
                const int64_t mine = lane < 16 ? s_tot[b][lane][v] : 0;
                int64_t x = mine;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    const int64_t y = __shfl_up(x, d, 64);
                    if (lane >= d) x += y;
                }
                if (lane < 16) s_pre[b][lane][v] = x - mine;
                if (lane == 15) s_tot[b][0][v] = x;  // the part's total, parked in slot 0
            }
    }
    __syncthreads();
    if (tid == 0) {
        int64_t nb = 0, lb = 0;
        for (int b = 0; b < p.n_parts; ++b) {
            s_base[b][0] = nb;
            s_base[b][1] = lb;
            nb += s_tot[b][0][1];
            lb += s_tot[b][0][2];
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < SPLIT_MAX_PARTS; ++b) {
        if (b >= p.n_parts) break;
        const int32_t *lc = p.leaves_cnt + (int64_t)b * M, *nc = p.nodes_cnt + (int64_t)b * M;
        const int64_t node_base = s_base[b][0], leaf_base = s_base[b][1];
        int64_t ea = s_pre[b][wave][0] + inc[b][0] - own[b][0];  // exclusive
        int64_t eb = s_pre[b][wave][1] + inc[b][1] - own[b][1];
        int64_t ec = s_pre[b][wave][2] + inc[b][2] - own[b][2];
        int64_t *cno = p.c_node_off + (int64_t)b * (M + 1), *cto = p.c_tree_off + (int64_t)b * (M + 1);
        for (int t = t0; t < t1; ++t) {
            const bool keep = lc[t] > 0;
            p.tree_pos[(int64_t)b * M + t] = keep ? (int32_t)ea : -1;
            p.node_start[(int64_t)b * M + t] = node_base + eb;
            p.leaf_start[(int64_t)b * M + t] = leaf_base + ec;
            if (keep) {
                cno[ea] = eb;  // relative to the child's own arrays
                cto[ea] = ec;
            }
            ea += keep;
            eb += nc[t];
            ec += lc[t];
        }
        if (tid == 1023) {
            const int64_t trees = s_tot[b][0][0], nodes = s_tot[b][0][1], leaves = s_tot[b][0][2];
            p.totals[b * 4 + 0] = trees;
            p.totals[b * 4 + 1] = nodes;
            p.totals[b * 4 + 2] = leaves;
            p.totals[b * 4 + 3] = node_base;
            cno[trees] = nodes;
            cto[trees] = leaves;
        }
    }
}

__global__ __launch_bounds__(SPLIT_THREADS) void k_split_fill(split_params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_fill[];
    double *l_len = (double *)s_fill, *l_sup = l_len + SPLIT_CAP, *l_cv = l_sup + SPLIT_CAP;
    int32_t *l_par = (int32_t *)(l_cv + SPLIT_CAP), *l_tax = l_par + SPLIT_CAP, *l_st = l_tax + SPLIT_CAP,
            *l_sv = l_st + SPLIT_CAP, *l_sj = l_sv + SPLIT_CAP, *l_cd = l_sj + SPLIT_CAP;
    unsigned char *l_mk = (unsigned char *)(l_cd + SPLIT_CAP);
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * p.tpb;
    const int t1 = min(p.n_trees, t0 + p.tpb);
    const int t = t0 + tid;
    const int np = p.n_parts;
    const int M = p.n_trees;
    const int64_t n0 = p.node_off[t0], n1 = p.node_off[t1];
    const bool stage = n1 - n0 <= SPLIT_CAP;
    const bool failed = p.flags[0] != 0;  // (uniform: written by the count kernel only)
    if (stage && !failed) {
        for (int i = tid; i < (int)(n1 - n0); i += SPLIT_THREADS) {
            l_par[i] = p.parent[n0 + i];
            l_tax[i] = p.taxon[n0 + i];
            l_len[i] = p.length[n0 + i];
            l_sup[i] = p.support[n0 + i];
            l_mk[i] = p.mark[n0 + i];
        }
        __syncthreads();
    }
    if (t >= t1 || failed) return;
    const int64_t off = p.node_off[t];
    const int32_t k = (int32_t)(p.node_off[t + 1] - off);
    const int32_t rel = (int32_t)(off - n0);
    const int32_t *par = stage ? l_par + rel : p.parent + off;
    const int32_t *tax = stage ? l_tax + rel : p.taxon + off;
    const double *len = stage ? l_len + rel : p.length + off;
    const double *sup = stage ? l_sup + rel : p.support + off;
    const unsigned char *mk = stage ? l_mk + rel : p.mark + off;
    int32_t *st = stage ? l_st + rel : p.stk_a + off;
    int32_t *sv = stage ? l_sv + rel : p.stk_b + off;
    int32_t *sj = stage ? l_sj + rel : p.stk_c + off;
    const double wt = p.weights[t];
    int64_t tree_base = 0;
    for (int b = 0; b < np; ++b) {
        const int32_t pos = p.tree_pos[(int64_t)b * M + t];
        const int64_t trees_b = p.totals[b * 4 + 0];
        if (pos < 0) {
            tree_base += trees_b;
            continue;
        }
        const int64_t base = p.node_start[(int64_t)b * M + t];
        int64_t slot = p.leaf_start[(int64_t)b * M + t];
        const int64_t slot_end = slot + p.leaves_cnt[(int64_t)b * M + t];
        // depth and value of the child's nodes by position (a tree's parts together hold at most k + 1 nodes)
        int32_t *cd = stage ? l_cd + rel : p.cdepth + base;
        double *cv = stage ? l_cv + rel : p.cval + base;
        p.c_weights[tree_base + pos] = wt;
        p.c_tree_index[tree_base + pos] = t;
        tree_base += trees_b;
        unsigned char *present = p.c_present + (int64_t)b * p.present_ld;
        int32_t sp = 0, vs = 0, j = 0;
        bool first_leaf = true;
        int32_t pend_depth = 0;
        double pend_val = 0.0;
        const unsigned bit = 1u << b;
        for (int32_t v = 0; v < k; ++v) {
            const int32_t q = par[v];
            while (sp > 0 && st[sp - 1] != q) {
                --sp;
                if (vs > 0 && sv[vs - 1] == st[sp]) --vs;  // a kept ancestor leaves the path
            }
            const int32_t x = tax[v];
            if (!(mk[v] & bit)) {
                if (x < 0) st[sp++] = v;
                continue;
            }
            const int32_t upj = vs > 0 ? sj[vs - 1] : -1;
            const int32_t up = vs > 0 ? sv[vs - 1] : -1;
            const int32_t ctx_ = x >= 0 ? p.new_id[x] : -1;
            double acc = len[v];
            for (int32_t u = v == 0 ? -1 : q; u >= 0 && u != up; u = par[u])
                if (!isnan(len[u]) && !isnan(acc)) acc = len[u] + acc;
            const double sp_ = sup[v];
            p.c_parent[base + j] = upj;
            p.c_taxon[base + j] = ctx_;
            p.c_length[base + j] = acc;
            p.c_support[base + j] = sp_;
            // ---- the child's tables, as scs_host_flatten walks the child in preorder
            if (j == 0) {
                cd[0] = 0;
                cv[0] = 0.0;
            } else {
                if (j != upj + 1) {  // not the first child: the next leaf's LCA with the previous one is upj
                    pend_depth = cd[upj];
                    pend_val = cv[upj];
                }
                if (x >= 0) {
                    if (!first_leaf) {
                        p.c_adj_depth[slot - 1] = pend_depth;
                        p.c_adj_val[slot - 1] = pend_val;
                    }
                    first_leaf = false;
                    if (slot < slot_end) p.c_leaf_taxon[slot] = ctx_;
                    ++slot;
                    present[ctx_] = 1;
                } else {
                    const double pv = cv[upj];
                    double val;
                    switch (p.strategy) {
                        case 0:
                            val = 1.0;
                            break;
                        case 1:
                            val = pv + 1.0;
                            break;
                        case 2:
                            val = pv + (isnan(acc) ? 1.0 : acc);
                            if (!isnan(acc) && acc < 0.0) p.flags[1 + b] = 0;
                            break;
                        default:
                            val = sp_;
                            if (isnan(val)) {
                                // (every inner node of a restricted tree has two or more children)
                                atomicExch(&p.flags[0], -3);
                                val = 0.0;
                            }
                            break;
                    }
                    cd[j] = cd[upj] + 1;
                    cv[j] = val;
                }
            }
            if (x < 0) {
                st[sp++] = v;
                sv[vs] = v;
                sj[vs] = j;
                ++vs;
            }
            ++j;
        }
        if (slot != slot_end) atomicExch(&p.flags[0], SCS_EINVAL);
        // padding slot so adj_* share the offsets of leaf_taxon
        p.c_adj_depth[slot_end - 1] = 0;
        p.c_adj_val[slot_end - 1] = 0.0;
    }
}


// ---------------------------------------------------------------------------
// Big trees: parallel WITHIN the tree (round 5).  One thread per tree leaves a forest of a few
// thousand trees of 10^5 nodes each to a few dozen waves that walk them alone (8-15 ns per node,
// slower than the host's team).  Every step of the restriction can be told per NODE instead:
//   * the previous leaf of the same part (exclusive max-scan of "my index if I am a leaf of part b");
//   * the LCA of two consecutive leaves x < y of a part = the first ancestor of y whose index is <= x
//     (walk up from y: consecutive leaves meet low in the tree) -- it is marked, as are the leaves;
//   * positions in the child = exclusive sum-scan of the marks; a kept node's parent in the child = its
//     first marked ancestor, the merged length folded on the way up (the chains of spliced nodes
//     are disjoint: linear work in all);
//   * depth and strategy value of a kept inner node from ITS OWN root path in the child (collected
//     walking up, summed top-down in the host's order of additions -- same bits; the paths of a
//     balanced tree are a few dozen nodes; a path longer than PAR_PATH sends the call back to the
//     one-thread-per-tree kernels);
//   * a leaf's table entries from the walk to the LCA with its predecessor.
// Launches over all nodes of the forest, coalesced where the tree allows; ~30 launches a split.
constexpr int PAR_PATH = 192;

enum { SCAN_SUM = 0, SCAN_MAX = 1 };

template <int OP>
__device__ __forceinline__ int32_t scan_op(int32_t a, int32_t b) {
    return OP == SCAN_SUM ? a + b : (a > b ? a : b);
}

// out[i] = op over in(j), j < i (exclusive), i in [0, n]; in(n) is never read.  4096 items per workgroup.
// (round 5, late: all parts of a split in ONE launch -- blockIdx.y is the part; `in0.with(part)` is the part's
// input, out and block_sums advance by a stride per part: a split of two parts made 24 scan launches, now 12)
template <int OP, typename F>
__global__ __launch_bounds__(256) void k_scan_local(F in0, int32_t *__restrict__ out, int64_t out_stride, int64_t n,
                                                     int32_t ident, int32_t *__restrict__ block_sums, int64_t nb) {
    __shared__ int32_t s_w[4];
    const F in = in0.with((int)blockIdx.y);
    out += (int64_t)blockIdx.y * out_stride;
    block_sums += (int64_t)blockIdx.y * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * 4096 + (int64_t)tid * 16;
    int32_t v[16];
    int32_t run = ident;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t i = base + k;
        const int32_t x = i < n ? in(i) : ident;
        v[k] = run;  // exclusive within the thread
        run = scan_op<OP>(run, x);
    }
    // exclusive scan of the threads' totals within the wave
    int32_t incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl = scan_op<OP>(o, incl);
    }
    int32_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = ident;
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int32_t wbase = ident;
    for (int w = 0; w < wave; ++w) wbase = scan_op<OP>(wbase, s_w[w]);
    const int32_t tbase = scan_op<OP>(wbase, excl);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t i = base + k;
        if (i <= n) out[i] = scan_op<OP>(tbase, v[k]);
    }
    if (tid == 255) block_sums[blockIdx.x] = scan_op<OP>(wbase, incl);
}

// exclusive scan of a part's block sums in place (one workgroup per part): shuffles inside the sixteen
// waves, their totals scanned by every thread from LDS -- two barriers per 1 024 blocks
template <int OP>
__global__ __launch_bounds__(1024) void k_scan_blocks(int32_t *__restrict__ block_sums, int64_t n_blocks, int32_t ident) {
    __shared__ int32_t s_w[16];
    __shared__ int32_t s_carry;
    block_sums += (int64_t)blockIdx.x * n_blocks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = ident;
    __syncthreads();
    for (int64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        const int64_t i = b0 + tid;
        const int32_t x = i < n_blocks ? block_sums[i] : ident;
        int32_t incl = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl = scan_op<OP>(o, incl);
        }
        int32_t excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = ident;
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int32_t wbase = s_carry;
        for (int w = 0; w < wave; ++w) wbase = scan_op<OP>(wbase, s_w[w]);
        if (i < n_blocks) block_sums[i] = scan_op<OP>(wbase, excl);
        __syncthreads();
        if (tid == 1023) s_carry = scan_op<OP>(wbase, incl);
        __syncthreads();
    }
}

template <int OP>
__global__ __launch_bounds__(256) void k_scan_add(int32_t *__restrict__ out, int64_t out_stride, int64_t n,
                                                   const int32_t *__restrict__ block_sums, int64_t nb) {
    out += (int64_t)blockIdx.y * out_stride;
    block_sums += (int64_t)blockIdx.y * nb;
    const int64_t i0 = (int64_t)blockIdx.x * 4096;
    const int32_t b = block_sums[blockIdx.x];
    for (int k = threadIdx.x; k < 4096; k += 256) {
        const int64_t i = i0 + k;
        if (i <= n) out[i] = scan_op<OP>(b, out[i]);
    }
}

// k_scan_blocks and k_scan_add in one launch while the blocks are few: every workgroup reduces the raw sums of the
// blocks in front of it itself (integers under + or max: any order gives the same value) and adds the result
template <int OP>
__global__ __launch_bounds__(256) void k_scan_add_raw(int32_t *__restrict__ out, int64_t out_stride, int64_t n,
                                                       const int32_t *__restrict__ block_sums, int64_t nb, int32_t ident) {
    __shared__ int32_t s_w[4];
    out += (int64_t)blockIdx.y * out_stride;
    block_sums += (int64_t)blockIdx.y * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t acc = ident;
    for (int64_t j = tid; j < (int64_t)blockIdx.x; j += 256) acc = scan_op<OP>(acc, block_sums[j]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc = scan_op<OP>(acc, __shfl_xor(acc, d, 64));
    if (lane == 0) s_w[wave] = acc;
    __syncthreads();
    const int32_t b = scan_op<OP>(scan_op<OP>(s_w[0], s_w[1]), scan_op<OP>(s_w[2], s_w[3]));
    const int64_t i0 = (int64_t)blockIdx.x * 4096;
    for (int k = tid; k < 4096; k += 256) {
        const int64_t i = i0 + k;
        if (i <= n) out[i] = scan_op<OP>(b, out[i]);
    }
}

// out[part][i], i in [0, n]: the exclusive scan of in.with(part) for every part (block_sums: n_parts x nb ints)
template <int OP, typename F>
static int scan_exclusive(F in, int32_t *out, int64_t out_stride, int n_parts, int64_t n, int32_t ident,
                          int32_t *block_sums, hipStream_t s) {
    const int64_t nb = (n + 1 + 4095) / 4096;  // (entry n = the total)
    k_scan_local<OP, F><<<dim3((unsigned)nb, (unsigned)n_parts), 256, 0, s>>>(in, out, out_stride, n, ident, block_sums, nb);
    if (nb > 1 && nb <= 8192) {
        k_scan_add_raw<OP><<<dim3((unsigned)nb, (unsigned)n_parts), 256, 0, s>>>(out, out_stride, n, block_sums, nb, ident);
    } else if (nb > 1) {
        k_scan_blocks<OP><<<(unsigned)n_parts, 1024, 0, s>>>(block_sums, nb, ident);
        k_scan_add<OP><<<dim3((unsigned)nb, (unsigned)n_parts), 256, 0, s>>>(out, out_stride, n, block_sums, nb);
    }
    SCS_HIP_CHECK(hipGetLastError());
    return SCS_OK;
}

struct par_params {
    int32_t n_trees, n_parts, strategy, n_taxa;
    int64_t n_nodes;
    const int64_t *node_off;
    const int32_t *parent, *taxon, *tree_id;
    const double *length, *support, *weights;
    const int32_t *part_of, *new_id;
    signed char *pc;        // [N] part of a leaf, -1
    int32_t *prev;          // [np][N + 1] previous leaf of the part (global index), -1
    int32_t *rank;          // [np][N + 1] leaves of the part before this node, in kept trees only (after the second scan)
    unsigned char *mark;    // [np][N]
    int32_t *kpos;          // [np][N + 1] position in the child (all parts' children: per part from 0)
    unsigned char *keep;    // [np][M]
    int32_t *corig;         // [NC] original node of a child position (per part at node_base)
    int32_t *cdepth;
    double *cval;
    // the same per-(part, tree) arrays and outputs as the serial path
    int32_t *leaves_cnt, *nodes_cnt, *tree_pos;
    int64_t *node_start, *leaf_start, *totals;
    int64_t *c_node_off, *c_tree_off;
    int32_t *c_parent, *c_taxon, *c_tree_index, *c_tree_id;
    double *c_length, *c_support, *c_weights;
    int32_t *c_leaf_taxon, *c_adj_depth;
    double *c_adj_val;
    unsigned char *c_present;
    int32_t present_ld;
    int32_t *flags;  // [0] error, [1 + b] monotone, [9] path overflow
};

struct f_key {
    const signed char *pc;
    int b;
    __device__ f_key with(int part) const { return f_key{pc, part}; }
    __device__ int32_t operator()(int64_t i) const { return pc[i] == b ? (int32_t)i : -1; }
};
struct f_ind {
    const signed char *pc;
    int b;
    __device__ f_ind with(int part) const { return f_ind{pc, part}; }
    __device__ int32_t operator()(int64_t i) const { return pc[i] == b ? 1 : 0; }
};
struct f_ind_kept {
    const signed char *pc;
    const unsigned char *keep;  // [parts][n_trees]
    const int32_t *tree_id;
    int64_t n_trees;
    int b;
    __device__ f_ind_kept with(int part) const { return f_ind_kept{pc, keep + part * n_trees, tree_id, n_trees, part}; }
    __device__ int32_t operator()(int64_t i) const { return (pc[i] == b && keep[tree_id[i]]) ? 1 : 0; }
};
struct f_mark {
    const unsigned char *mark;  // [parts][n_nodes]
    int64_t n_nodes;
    __device__ f_mark with(int part) const { return f_mark{mark + part * n_nodes, n_nodes}; }
    __device__ int32_t operator()(int64_t i) const { return mark[i]; }
};

__global__ void k_par_tree_id(const int64_t *__restrict__ node_off, int32_t n_trees, int64_t n, int32_t *__restrict__ tree_id) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t lo = 0, hi = n_trees - 1;  // last t with node_off[t] <= i
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (node_off[mid] <= i) lo = mid;
        else hi = mid - 1;
    }
    tree_id[i] = lo;
}

// scs_forest_upload's argument check on the uploaded copy: every later kernel walks `parent` upwards and
// indexes by `taxon` without looking again (a non-root without a smaller parent would walk out of its tree
// or never stop)
__global__ void k_validate_forest(const int64_t *__restrict__ node_off, const int32_t *__restrict__ parent,
                                  const int32_t *__restrict__ taxon, int32_t n_trees, int32_t n_taxa, int64_t n,
                                  int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t lo = 0, hi = n_trees - 1;  // last t with node_off[t] <= i
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (node_off[mid] <= i) lo = mid;
        else hi = mid - 1;
    }
    const int64_t rel = i - node_off[lo];
    const int32_t q = parent[i], x = taxon[i];
    const bool ok = (rel == 0 ? q == -1 : (q >= 0 && q < rel)) && x >= -1 && x < n_taxa;
    if (!ok) atomicExch(flag, 1);
}

__global__ void k_par_pc(par_params p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_nodes) return;
    const int32_t x = p.taxon[i];
    p.pc[i] = x >= 0 ? (signed char)p.part_of[x] : (signed char)-1;
    const int64_t t0 = p.node_off[p.tree_id[i]];
    const int32_t q = p.parent[i];
    if (i > t0 && (q < 0 || q >= i - t0)) atomicExch(&p.flags[0], SCS_EINVAL);  // not preorder
}

// per (part, tree): leaves of the part, kept or dropped
__global__ void k_par_keep(par_params p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_trees) return;
    for (int b = 0; b < p.n_parts; ++b) {
        const int32_t *r = p.rank + (int64_t)b * (p.n_nodes + 1);
        const int32_t c = r[p.node_off[t + 1]] - r[p.node_off[t]];
        p.keep[(int64_t)b * p.n_trees + t] = c >= 2;
        p.leaves_cnt[(int64_t)b * p.n_trees + t] = c >= 2 ? c : 0;
    }
}

// every leaf of a part marks itself and the LCA with the part's previous leaf in the tree
__global__ void k_par_mark(par_params p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_nodes) return;
    const int b = p.pc[i];
    if (b < 0) return;
    const int32_t t = p.tree_id[i];
    if (!p.keep[(int64_t)b * p.n_trees + t]) return;
    unsigned char *mk = p.mark + (int64_t)b * p.n_nodes;
    mk[i] = 1;
    const int64_t t0 = p.node_off[t];
    const int64_t x = p.prev[(int64_t)b * (p.n_nodes + 1) + i];
    if (x < t0) return;  // the part's first leaf in this tree
    int64_t u = t0 + p.parent[i];
    while (u > x) u = t0 + p.parent[u];
    mk[u] = 1;
}

// per (part, tree): nodes of the child tree
__global__ void k_par_nodes(par_params p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_trees) return;
    for (int b = 0; b < p.n_parts; ++b) {
        const int32_t *kp = p.kpos + (int64_t)b * (p.n_nodes + 1);
        p.nodes_cnt[(int64_t)b * p.n_trees + t] = kp[p.node_off[t + 1]] - kp[p.node_off[t]];
    }
}

// every kept node: its parent in the child, the merged length, its place
__global__ void k_par_nodes_fill(par_params p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_nodes) return;
    const int32_t t = p.tree_id[i];
    const int64_t t0 = p.node_off[t];
    const int32_t x = p.taxon[i];
    for (int b = 0; b < p.n_parts; ++b) {
        const unsigned char *mk = p.mark + (int64_t)b * p.n_nodes;
        if (!mk[i]) continue;
        const int32_t *kp = p.kpos + (int64_t)b * (p.n_nodes + 1);
        const int64_t node_base = p.totals[b * 4 + 3];
        const int64_t j = node_base + kp[i];
        double acc = p.length[i];
        int64_t u = i == t0 ? -1 : t0 + p.parent[i];
        while (u >= 0 && !mk[u]) {
            const double lu = p.length[u];
            if (!isnan(lu) && !isnan(acc)) acc = lu + acc;
            u = u == t0 ? -1 : t0 + p.parent[u];
        }
        p.c_parent[j] = u < 0 ? -1 : kp[u] - kp[t0];
        p.c_taxon[j] = x >= 0 ? p.new_id[x] : -1;
        p.c_length[j] = acc;
        p.c_support[j] = p.support[i];
        p.c_tree_id[j] = p.tree_pos[(int64_t)b * p.n_trees + t];
        p.corig[j] = (int32_t)(i - t0);
        if (x >= 0) p.c_present[(int64_t)b * p.present_ld + p.new_id[x]] = 1;
    }
}

// every kept inner node: depth and strategy value from its own root path in the child
__global__ void k_par_values(par_params p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_nodes) return;
    if (p.taxon[i] >= 0) return;
    const int32_t t = p.tree_id[i];
    const int64_t t0 = p.node_off[t];
    for (int b = 0; b < p.n_parts; ++b) {
        if (!p.mark[(int64_t)b * p.n_nodes + i]) continue;
        const int32_t *kp = p.kpos + (int64_t)b * (p.n_nodes + 1);
        const int64_t node_base = p.totals[b * 4 + 3];
        const int64_t c0 = node_base + kp[t0];  // the child tree's root
        const int64_t j = node_base + kp[i];
        double path[PAR_PATH];
        int d = 0;
        bool overflow = false;
        for (int64_t q = j; q != c0;) {
            if (d == PAR_PATH) {
                overflow = true;
                break;
            }
            path[d++] = p.c_length[q];
            q = c0 + p.c_parent[q];
        }
        if (overflow) {
            atomicExch(&p.flags[9], 1);
            continue;
        }
        double val = 0.0;
        switch (p.strategy) {
            case 0:
                val = d > 0 ? 1.0 : 0.0;
                break;
            case 1:
                for (int k = 0; k < d; ++k) val = val + 1.0;
                break;
            case 2:
                for (int k = d - 1; k >= 0; --k) {
                    const double l = path[k];
                    val = val + (isnan(l) ? 1.0 : l);
                    if (!isnan(l) && l < 0.0) p.flags[1 + b] = 0;
                }
                break;
            default:
                val = d > 0 ? p.c_support[j] : 0.0;
                if (d > 0 && isnan(val)) {
                    atomicExch(&p.flags[0], -3);
                    val = 0.0;
                }
                break;
        }
        p.cdepth[j] = d;
        p.cval[j] = val;
    }
}

// every kept leaf: its slot, and the table entry between its predecessor and itself
__global__ void k_par_tables(par_params p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_nodes) return;
    const int b = p.pc[i];
    if (b < 0) return;
    const int32_t t = p.tree_id[i];
    if (!p.keep[(int64_t)b * p.n_trees + t]) return;
    const int64_t t0 = p.node_off[t], t1 = p.node_off[t + 1];
    const int32_t *r = p.rank + (int64_t)b * (p.n_nodes + 1);
    const int32_t *kp = p.kpos + (int64_t)b * (p.n_nodes + 1);
    int64_t leaf_base = 0;
    for (int c = 0; c < b; ++c) leaf_base += p.totals[c * 4 + 2];
    const int64_t slot = leaf_base + r[i];
    p.c_leaf_taxon[slot] = p.new_id[p.taxon[i]];
    if (r[i] + 1 == r[t1]) {  // the tree's last leaf of the part: the padding entry
        p.c_adj_depth[slot] = 0;
        p.c_adj_val[slot] = 0.0;
    }
    const int64_t x = p.prev[(int64_t)b * (p.n_nodes + 1) + i];
    if (x < t0) return;
    const int64_t node_base = p.totals[b * 4 + 3];
    const int64_t c0 = node_base + kp[t0];
    int64_t q = c0 + p.c_parent[node_base + kp[i]];
    while (p.corig[q] > (int32_t)(x - t0)) q = c0 + p.c_parent[q];
    p.c_adj_depth[slot - 1] = p.cdepth[q];
    p.c_adj_val[slot - 1] = p.cval[q];
}

// per (part, kept tree): weights and the index in the parent (offsets come from k_split_scan)
__global__ void k_par_trees(par_params p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n_trees) return;
    int64_t tree_base = 0;
    for (int b = 0; b < p.n_parts; ++b) {
        const int32_t pos = p.tree_pos[(int64_t)b * p.n_trees + t];
        if (pos >= 0) {
            p.c_weights[tree_base + pos] = p.weights[t];
            p.c_tree_index[tree_base + pos] = t;
        }
        tree_base += p.totals[b * 4 + 0];
    }
}


// ---------------------------------------------------------------------------
// Level mode (round 6): ONE split for a whole level of the recursion.
//
// The nodes of one level of the recursion below some node hold disjoint taxon sets, so their forests can
// live in ONE forest -- the trees of node 0, then of node 1, ... -- over one numbering of the taxa (a
// "universe": node k owns a consecutive id range), and one call restricts all of them to all of their
// parts: part_of[x] is the part of taxon x INSIDE ITS NODE (0 .. 7), new_id[x] its id in the next level's
// universe.  The children come back as ONE forest again (part-major: the first parts of all nodes, then the
// second parts, ...; inside a part the parent's tree order, i.e. node by node), with its tables, and the
// host learns only what it needs to go on: trees and leaves per child, present taxa, the connected
// components of the proper cluster graph of every child and a 128-bit signature per taxon whose equality
// is NECESSARY for two taxa to be contracted (reference: scs.py:122 `_get_graph_components`, :302-316 the
// contraction relation; restriction :411-455).  What used to be one scs_forest_split + one download of
// 16 bytes per leaf + a host union-find per NODE is one call per LEVEL.
//
// k_split_scan is one workgroup; a level forest has millions of trees: the three prefix sums run on the
// multi-block scans of the node-parallel family instead.
struct f_cnt_pos {
    const int32_t *a;
    int64_t stride;
    __device__ f_cnt_pos with(int part) const { return f_cnt_pos{a + (int64_t)part * stride, stride}; }
    __device__ int32_t operator()(int64_t i) const { return a[i] > 0 ? 1 : 0; }
};
struct f_cnt {
    const int32_t *a;
    int64_t stride;
    __device__ f_cnt with(int part) const { return f_cnt{a + (int64_t)part * stride, stride}; }
    __device__ int32_t operator()(int64_t i) const { return a[i]; }
};

// sc_*: [n_parts][M + 1] exclusive sums (entry M = the part's total) of kept trees, nodes, leaves
__global__ __launch_bounds__(256) void k_split_finalize(split_params p, const int32_t *__restrict__ sc_keep,
                                                        const int32_t *__restrict__ sc_nodes,
                                                        const int32_t *__restrict__ sc_leaves) {
    const int M = p.n_trees;
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    int64_t node_base = 0, leaf_base = 0;
    for (int c = 0; c < b; ++c) {
        node_base += sc_nodes[(int64_t)c * (M + 1) + M];
        leaf_base += sc_leaves[(int64_t)c * (M + 1) + M];
    }
    const int64_t row = (int64_t)b * (M + 1);
    const int32_t ea = sc_keep[row + t], eb = sc_nodes[row + t], ec = sc_leaves[row + t];
    const bool keep = p.leaves_cnt[(int64_t)b * M + t] > 0;
    p.tree_pos[(int64_t)b * M + t] = keep ? ea : -1;
    p.node_start[(int64_t)b * M + t] = node_base + eb;
    p.leaf_start[(int64_t)b * M + t] = leaf_base + ec;
    int64_t *cno = p.c_node_off + row, *cto = p.c_tree_off + row;
    if (keep) {
        cno[ea] = eb;
        cto[ea] = ec;
    }
    if (t == M - 1) {
        const int64_t trees = sc_keep[row + M], nodes = sc_nodes[row + M], leaves = sc_leaves[row + M];
        p.totals[b * 4 + 0] = trees;
        p.totals[b * 4 + 1] = nodes;
        p.totals[b * 4 + 2] = leaves;
        p.totals[b * 4 + 3] = node_base;
        cno[trees] = nodes;
        cto[trees] = leaves;
    }
}

// offsets of the union of all parts' children (tree order: part-major)
__global__ __launch_bounds__(256) void k_level_union(split_params p, int64_t *__restrict__ u_node_off,
                                                     int64_t *__restrict__ u_tree_off) {
    const int M = p.n_trees;
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    int64_t tree_base = 0, leaf_base = 0;
    for (int c = 0; c < b; ++c) {
        tree_base += p.totals[c * 4 + 0];
        leaf_base += p.totals[c * 4 + 2];
    }
    const int64_t node_base = p.totals[b * 4 + 3];
    const int32_t pos = p.tree_pos[(int64_t)b * M + t];
    if (pos >= 0) {
        const int64_t row = (int64_t)b * (M + 1);
        u_node_off[tree_base + pos] = node_base + p.c_node_off[row + pos];
        u_tree_off[tree_base + pos] = leaf_base + p.c_tree_off[row + pos];
    }
    if (t == M - 1 && b == p.n_parts - 1) {
        const int64_t trees = tree_base + p.totals[b * 4 + 0];
        u_node_off[trees] = node_base + p.totals[b * 4 + 1];
        u_tree_off[trees] = leaf_base + p.totals[b * 4 + 2];
    }
}

// trees and leaves of child (part b, node k): the part's kept trees carry their index in the parent forest
// in increasing order (c_tree_index), so a node's share is a range of them -- two binary searches, no atomics
__global__ void k_level_counts(split_params p, int32_t n_nodes, const int32_t *__restrict__ node_tree_end,
                               int32_t *__restrict__ child_trees, int64_t *__restrict__ child_leaves) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.n_parts * n_nodes) return;
    const int b = (int)(i / n_nodes), k = (int)(i - (int64_t)b * n_nodes);
    int64_t tree_base = 0;
    for (int c = 0; c < b; ++c) tree_base += p.totals[c * 4 + 0];
    const int32_t trees = (int32_t)p.totals[b * 4 + 0];
    const int32_t *idx = p.c_tree_index + tree_base;
    const int32_t t_lo = k ? node_tree_end[k - 1] : 0, t_hi = node_tree_end[k];
    int32_t pos[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int32_t want = e ? t_hi : t_lo;  // first kept tree with index >= want
        int32_t lo = 0, hi = trees;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (idx[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        pos[e] = lo;
    }
    const int64_t *cto = p.c_tree_off + (int64_t)b * (p.n_trees + 1);
    child_trees[i] = pos[1] - pos[0];
    child_leaves[i] = cto[pos[1]] - cto[pos[0]];
}

// ---- analysis of a forest's tables: components and contraction signatures ------------------
// Two taxa are adjacent in the proper cluster graph iff they share a root side in some tree
// (scs.py:651-652, whatever the weight), so the components are those of "join consecutive leaves of a side":
// leaf p and p + 1 unless the gap between them is a root gap (adj_depth 0; the padding slot behind a
// tree's last leaf is 0 too).  Lock-free union-find, the larger root hooked under the smaller: a set's
// root is its smallest member -- the order the host numbers components in.
__device__ __forceinline__ int32_t uf_find(int32_t *parent, int32_t x) {
    int32_t q = parent[x];
    while (q != x) {
        const int32_t g = parent[q];
        if (g != q) parent[x] = g;  // (path halving; a benign race: only ever towards an ancestor)
        x = q;
        q = g;
    }
    return x;
}

__global__ void k_uf_init(int32_t *parent, int32_t n, unsigned long long *sig) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        parent[i] = i;
        sig[2 * i] = 0;
        sig[2 * i + 1] = 0;
    }
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct f_side_start {
    const int32_t *adj_depth;
    const int64_t *n_leaves;  // (device: the union's leaf count is only known there)
    __device__ f_side_start with(int) const { return *this; }
    __device__ int32_t operator()(int64_t i) const {
        if (i >= *n_leaves) return -1;
        return (i == 0 || adj_depth[i - 1] == 0) ? (int32_t)i : -1;
    }
};

__global__ void k_leaf_total(const int64_t *totals, int n_parts, int64_t *out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t l = 0;
        for (int b = 0; b < n_parts; ++b) l += totals[b * 4 + 2];
        *out = l;
    }
}

// per leaf: join with the next leaf of the side; add the side's key to the taxon's signature.  A taxon's
// signature is the SET of (tree, root side) it occurs in (flatten.contraction_groups: two taxa are contracted
// iff their sets are equal); the sum of a 2 x 64-bit mix of the side's first leaf slot over that set is equal
// for equal sets -- so distinct sums PROVE that nothing is contracted, and equal sums send the node to the
// exact host routine.
// (measured and not kept, round 6: several leaves per thread with all their loads in flight, and "two leaves with
// the same parent pointer are in one set" before any find -- 2.7 -> 2.5 ms a level at 20 000 taxa x 5 000 trees:
// what this kernel waits for is the hooking of the first trees of every node, thousands of threads on a few roots)
template <bool SIG>
__global__ void k_analyze_leaves(const int32_t *__restrict__ leaf_taxon, const int32_t *__restrict__ adj_depth,
                                 const int32_t *__restrict__ side_excl, const int64_t *__restrict__ n_leaves,
                                 int32_t *parent, unsigned long long *sig, int sample = 1) {
    // sample > 1 (a first pass): this wave takes the leaves of one wave in `sample`
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = ((g >> 6) * sample << 6) | (g & 63);
    if (p >= *n_leaves) return;
    const int32_t x = leaf_taxon[p];
    if (SIG) {
        const bool start = p == 0 || adj_depth[p - 1] == 0;
        const unsigned long long side = (unsigned long long)(start ? (int32_t)p : side_excl[p]);
        // (one 64-bit sum, the second word stays 0: DISTINCT sums prove distinct sets whatever the width, and a
        // false collision -- ~T^2 / 2^65 a level -- only sends a node to the exact routine for nothing)
        atomicAdd(&sig[2 * x], mix64(side * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull));
    }
    if (adj_depth[p] == 0) return;  // a root gap, or the tree's last leaf
    int32_t u = x, v = leaf_taxon[p + 1];
    for (;;) {
        u = uf_find(parent, u);
        v = uf_find(parent, v);
        if (u == v) break;
        if (u < v) {
            const int32_t w = u;
            u = v;
            v = w;
        }
        const int32_t old = atomicCAS(&parent[u], u, v);  // hook the larger root under the smaller
        if (old == u) break;
        u = old;
    }
}

// The signatures of a universe above ANALYZE_LDS_TAXA ids: the global atomic per leaf was a quarter to a third of the
// kernel above (3.96 ms a level with two 64-bit adds per leaf, 2.99 with one, 2.72 with none, at 20 000 taxa x
// 5 000 trees).  Here a workgroup folds a long run of leaves into the partial sums of ONE tile
// of SIG_TILE taxa in LDS (LDS atomics; the other tiles' leaves are skipped -- the run is read once per tile,
// coalesced) and publishes one add per taxon it met: 50 x fewer global atomics.  Sums mod 2^64: any order.
constexpr int ANALYZE_SAMPLE = 256;  // the union-find's first pass: one wave of leaves in this many
constexpr int SIG_TILE = 16384;  // x 8 bytes = 128 KB of the workgroup's LDS
__global__ __launch_bounds__(1024) void k_analyze_sig_tiled(const int32_t *__restrict__ leaf_taxon,
                                                            const int32_t *__restrict__ adj_depth,
                                                            const int32_t *__restrict__ side_excl,
                                                            const int64_t *__restrict__ n_leaves, int64_t chunk,
                                                            int32_t n_taxa, unsigned long long *sig) {
    extern __shared__ unsigned long long l_tile[];
    const int64_t L = *n_leaves;
    const int64_t p0 = (int64_t)blockIdx.x * chunk;
    if (p0 >= L) return;
    const int64_t p1 = p0 + chunk < L ? p0 + chunk : L;
    const int32_t lo = (int32_t)blockIdx.y * SIG_TILE;
    const int32_t n = n_taxa - lo < SIG_TILE ? n_taxa - lo : SIG_TILE;
    for (int i = threadIdx.x; i < n; i += 1024) l_tile[i] = 0;
    __syncthreads();
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 1024) {
        const int32_t x = leaf_taxon[p] - lo;
        if ((uint32_t)x >= (uint32_t)n) continue;
        const bool start = p == 0 || adj_depth[p - 1] == 0;
        const unsigned long long side = (unsigned long long)(start ? (int32_t)p : side_excl[p]);
        atomicAdd(&l_tile[x], mix64(side * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const unsigned long long v = l_tile[i];
        if (v) atomicAdd(&sig[2 * (lo + i)], v);
    }
}

// The same for a universe of at most ANALYZE_LDS_TAXA ids (every level below a node of that size): thousands of
// trees over a few hundred taxa make every leaf's two signature adds -- and the union-find's hooks -- land on the
// same few addresses (measured: 470 us a level, the atomics serialised in the L2).  A workgroup folds a run of
// leaves in LDS first (its own union-find and partial signatures) and publishes one add per taxon it met and
// one union per taxon it hooked.  Sums mod 2^64 and set unions: the order does not matter.
constexpr int ANALYZE_LDS_TAXA = 2048;
constexpr int ANALYZE_LEAVES_PER_BLOCK = 256 * 32;

__device__ __forceinline__ void uf_union(int32_t *parent, int32_t u, int32_t v) {
    for (;;) {
        u = uf_find(parent, u);
        v = uf_find(parent, v);
        if (u == v) return;
        if (u < v) {
            const int32_t w = u;
            u = v;
            v = w;
        }
        const int32_t old = atomicCAS(&parent[u], u, v);  // hook the larger root under the smaller
        if (old == u) return;
        u = old;
    }
}

__global__ __launch_bounds__(256) void k_analyze_leaves_lds(const int32_t *__restrict__ leaf_taxon,
                                                            const int32_t *__restrict__ adj_depth,
                                                            const int32_t *__restrict__ side_excl,
                                                            const int64_t *__restrict__ n_leaves, int32_t n_taxa,
                                                            int32_t *parent, unsigned long long *sig) {
    __shared__ int32_t l_par[ANALYZE_LDS_TAXA];
    __shared__ unsigned long long l_sig[2 * ANALYZE_LDS_TAXA];
    const int64_t L = *n_leaves;
    const int64_t p0 = (int64_t)blockIdx.x * ANALYZE_LEAVES_PER_BLOCK;
    if (p0 >= L) return;
    for (int x = threadIdx.x; x < n_taxa; x += 256) {
        l_par[x] = x;
        l_sig[2 * x] = 0;
        l_sig[2 * x + 1] = 0;
    }
    __syncthreads();
    const int64_t p1 = p0 + ANALYZE_LEAVES_PER_BLOCK < L ? p0 + ANALYZE_LEAVES_PER_BLOCK : L;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        const int32_t x = leaf_taxon[p];
        const bool start = p == 0 || adj_depth[p - 1] == 0;
        const unsigned long long side = (unsigned long long)(start ? (int32_t)p : side_excl[p]);
        atomicAdd(&l_sig[2 * x], mix64(side * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull));
        atomicAdd(&l_sig[2 * x + 1], mix64((side + 0x2545F4914F6CDD1Dull) * 0xD1342543DE82EF95ull));
        if (adj_depth[p] != 0) uf_union(l_par, x, leaf_taxon[p + 1]);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < n_taxa; x += 256) {
        const unsigned long long a = l_sig[2 * x], b = l_sig[2 * x + 1];
        if (a | b) {  // (a taxon this run of leaves never met adds nothing; a met one adds its mixes, never 0 | 0 in practice)
            atomicAdd(&sig[2 * x], a);
            atomicAdd(&sig[2 * x + 1], b);
        }
        const int32_t q = l_par[x];
        if (q != x) uf_union(parent, x, q);
    }
}

__global__ void k_uf_flatten(int32_t *parent, int32_t n, int32_t *root) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) root[i] = uf_find(parent, i);
}

// components and signatures of the tables [0, *d_n_leaves) over n_taxa ids: comp_root / sig are device
// outputs ([n_taxa] / [n_taxa][2]); scratch from `alloc`
template <typename A>
static int analyze_tables(A &alloc, const int32_t *leaf_taxon, const int32_t *adj_depth, const int64_t *d_n_leaves,
                          int64_t leaf_cap, int32_t n_taxa, int32_t *comp_root, unsigned long long *sig, hipStream_t s,
                          int max_lds_bytes) {
    int32_t *parent = nullptr, *side = nullptr, *block_sums = nullptr;
    SCS_TRY(alloc((size_t)n_taxa * 4, (void **)&parent));
    SCS_TRY(alloc((size_t)(leaf_cap + 1) * 4, (void **)&side));
    SCS_TRY(alloc((size_t)((leaf_cap + 1 + 4095) / 4096 + 1) * 4, (void **)&block_sums));
    k_uf_init<<<(unsigned)((n_taxa + 255) / 256), 256, 0, s>>>(parent, n_taxa, sig);
    SCS_TRY((scan_exclusive<SCAN_MAX>(f_side_start{adj_depth, d_n_leaves}, side, leaf_cap + 1, 1, leaf_cap, -1,
                                      block_sums, s)));
    if (leaf_cap > 0 && n_taxa <= ANALYZE_LDS_TAXA)
        k_analyze_leaves_lds<<<(unsigned)((leaf_cap + ANALYZE_LEAVES_PER_BLOCK - 1) / ANALYZE_LEAVES_PER_BLOCK), 256, 0, s>>>(
            leaf_taxon, adj_depth, side, d_n_leaves, n_taxa, parent, sig);
    else if (leaf_cap > 0 && max_lds_bytes >= SIG_TILE * 8 && !scs_dbg("SCS_ANALYZE_GLOBAL_SIG")) {
        static bool attr_set = false;
        if (!attr_set) {
            SCS_HIP_CHECK(hipFuncSetAttribute((const void *)k_analyze_sig_tiled, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              SIG_TILE * 8));
            attr_set = true;
        }
        // Two passes (round 6).  Launched over all leaves at once, the union-find spends its time BUILDING the sets:
        // 10^8 threads arrive at singletons together and hook and re-hook a few roots (2.7 ms a level at 20 000
        // taxa x 5 000 trees; the same launch on the finished structure: 65 us).  A first pass over one wave of
        // leaves in ANALYZE_SAMPLE builds nearly all of it with 1/256 of the contenders; the full pass then finds
        // nearly every pair joined already (0.19 + 0.09 ms).  Unions are idempotent and order-free: same sets, and
        // the root of a set is its smallest member either way.
        int sample = ANALYZE_SAMPLE;
        if (const char *e = scs_dbg("SCS_ANALYZE_SAMPLE")) sample = atoi(e) > 1 ? atoi(e) : 1;
        // (a coarser pass in front of it where there are leaves enough: 13 us, and the pass after it meets fewer singletons)
        for (int smp : {sample * 16, sample}) {
            if (smp <= 1 || leaf_cap <= (int64_t)64 * smp * 4) continue;
            const int64_t waves = (leaf_cap + 63) / 64, some = (waves + smp - 1) / smp;
            k_analyze_leaves<false><<<(unsigned)((some * 64 + 255) / 256), 256, 0, s>>>(leaf_taxon, adj_depth, side,
                                                                                        d_n_leaves, parent, sig, smp);
        }
        k_analyze_leaves<false><<<(unsigned)((leaf_cap + 255) / 256), 256, 0, s>>>(leaf_taxon, adj_depth, side,
                                                                                   d_n_leaves, parent, sig);
        // runs of leaves: about two workgroups per CU and tile, at least 16 384 leaves each
        int64_t chunk = (leaf_cap + 511) / 512;
        chunk = (chunk + 1023) / 1024 * 1024;
        if (chunk < 16384) chunk = 16384;
        const dim3 grid((unsigned)((leaf_cap + chunk - 1) / chunk), (unsigned)((n_taxa + SIG_TILE - 1) / SIG_TILE));
        k_analyze_sig_tiled<<<grid, 1024, SIG_TILE * 8, s>>>(leaf_taxon, adj_depth, side, d_n_leaves, chunk, n_taxa, sig);
    } else if (leaf_cap > 0)
        k_analyze_leaves<true><<<(unsigned)((leaf_cap + 255) / 256), 256, 0, s>>>(leaf_taxon, adj_depth, side,
                                                                                  d_n_leaves, parent, sig);
    k_uf_flatten<<<(unsigned)((n_taxa + 255) / 256), 256, 0, s>>>(parent, n_taxa, comp_root);
    SCS_HIP_CHECK(hipGetLastError());
    return SCS_OK;
}

}  // namespace

extern "C" int scs_forest_upload(scs_ctx *ctx, int32_t n_taxa, int32_t n_trees, const int64_t *node_off,
                                 const int32_t *parent, const int32_t *taxon, const double *length,
                                 const double *support, const double *weights, int64_t n_leaves,
                                 scs_forest **out) {
    SCS_REQUIRE(ctx && node_off && parent && taxon && length && support && weights && out,
                "scs_forest_upload: null argument");
    SCS_REQUIRE(n_taxa >= 1 && n_trees >= 1, "scs_forest_upload: need at least one taxon and one tree");
    SCS_REQUIRE(node_off[0] == 0, "scs_forest_upload: node_off must start at 0");
    for (int32_t t = 0; t < n_trees; ++t)
        SCS_REQUIRE(node_off[t + 1] > node_off[t] && node_off[t + 1] - node_off[t] <= INT32_MAX / 4,
                    "scs_forest_upload: tree %d has a bad node count", t);
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t N = node_off[n_trees];
    auto f = std::unique_ptr<scs_forest>(new scs_forest());
    f->n_taxa = n_taxa;
    f->n_trees = n_trees;
    f->n_nodes = N;
    f->region = std::make_shared<forest_region>();
    f->region->ctx = ctx;
    SCS_TRY(f->region->alloc((size_t)(n_trees + 1) * 8, (void **)&f->node_off));
    SCS_TRY(f->region->alloc((size_t)N * 4, (void **)&f->parent));
    SCS_TRY(f->region->alloc((size_t)N * 4, (void **)&f->taxon));
    SCS_TRY(f->region->alloc((size_t)N * 8, (void **)&f->length));
    SCS_TRY(f->region->alloc((size_t)N * 8, (void **)&f->support));
    SCS_TRY(f->region->alloc((size_t)n_trees * 8, (void **)&f->weights));
    hipStream_t s = ctx->stream;
    SCS_HIP_CHECK(hipMemcpyAsync(f->node_off, node_off, (size_t)(n_trees + 1) * 8, hipMemcpyHostToDevice, s));
    SCS_HIP_CHECK(hipMemcpyAsync(f->parent, parent, (size_t)N * 4, hipMemcpyHostToDevice, s));
    SCS_HIP_CHECK(hipMemcpyAsync(f->taxon, taxon, (size_t)N * 4, hipMemcpyHostToDevice, s));
    SCS_HIP_CHECK(hipMemcpyAsync(f->length, length, (size_t)N * 8, hipMemcpyHostToDevice, s));
    SCS_HIP_CHECK(hipMemcpyAsync(f->support, support, (size_t)N * 8, hipMemcpyHostToDevice, s));
    SCS_HIP_CHECK(hipMemcpyAsync(f->weights, weights, (size_t)n_trees * 8, hipMemcpyHostToDevice, s));
    // the arrays are checked where they now are (preorder parents, taxon ids in range)
    int32_t *d_flag = nullptr;
    SCS_TRY(scs_block_alloc(ctx, 64, (void **)&d_flag));
    struct flag_guard {
        scs_ctx *ctx;
        void *p;
        ~flag_guard() { scs_block_release(ctx, p); }
    } fg{ctx, d_flag};
    SCS_HIP_CHECK(hipMemsetAsync(d_flag, 0, 4, s));
    k_validate_forest<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(f->node_off, f->parent, f->taxon, n_trees, n_taxa, N, d_flag);
    SCS_HIP_CHECK(hipGetLastError());
    int32_t bad = 0;
    SCS_HIP_CHECK(hipMemcpyAsync(&bad, d_flag, 4, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));  // (the caller's arrays may go away)
    SCS_REQUIRE(bad == 0, "scs_forest_upload: malformed tree arrays (a parent that is not an earlier node of the "
                          "tree, a root with a parent, or a taxon id outside [-1, n_taxa))");
    SCS_REQUIRE(n_leaves >= n_trees && n_leaves <= N, "scs_forest_upload: bad leaf count");
    f->n_leaves = n_leaves;  // (the number of nodes with taxon >= 0: sizes the children's arrays)
    *out = f.release();
    return SCS_OK;
}

extern "C" int scs_forest_free(scs_ctx *ctx, scs_forest *f) {
    (void)ctx;
    delete f;  // (the region goes with its last forest)
    return SCS_OK;
}

// level mode (scs_forest_split_level): the nodes of the level as consecutive tree ranges of `f`, and what
// the host wants back per child and per taxon of the children's universe
struct level_args {
    int32_t n_nodes = 0;
    const int32_t *node_tree_end = nullptr;  // host [n_nodes]: exclusive end of node k's trees (the last = n_trees)
    int32_t child_taxa = 0;                  // size of the children's universe (new_id < child_taxa)
    int32_t *child_trees = nullptr;          // host out [n_parts][n_nodes]
    int64_t *child_leaves = nullptr;         // host out [n_parts][n_nodes]
    uint8_t *present = nullptr;              // host out [child_taxa]
    int32_t *comp_root = nullptr;            // host out [child_taxa]
    uint64_t *sig = nullptr;                 // host out [child_taxa][2]
};

static int forest_split(scs_ctx *ctx, const scs_forest *f, const int32_t *part_of, const int32_t *new_id,
                        int32_t n_parts, const int32_t *part_taxa, int32_t strategy, scs_forest **out_forests,
                        scs_forest_info *info, bool force_serial, const level_args *lv = nullptr);

extern "C" int scs_forest_split(scs_ctx *ctx, const scs_forest *f, const int32_t *part_of, const int32_t *new_id,
                                int32_t n_parts, const int32_t *part_taxa, int32_t strategy,
                                scs_forest **out_forests, scs_forest_info *info) {
    SCS_REQUIRE(part_taxa != nullptr, "scs_forest_split: null argument");
    return forest_split(ctx, f, part_of, new_id, n_parts, part_taxa, strategy, out_forests, info, false);
}

extern "C" int scs_forest_split_level(scs_ctx *ctx, const scs_forest *f, const int32_t *part_of,
                                      const int32_t *new_id, int32_t n_parts, int32_t child_taxa, int32_t strategy,
                                      int32_t n_nodes, const int32_t *node_tree_end, scs_forest **out_union,
                                      scs_forest_info *info, int32_t *child_trees, int64_t *child_leaves,
                                      uint8_t *present, int32_t *comp_root, uint64_t *sig) {
    SCS_REQUIRE(ctx && f && node_tree_end && out_union && info && child_trees && child_leaves && present &&
                    comp_root && sig,
                "scs_forest_split_level: null argument");
    SCS_REQUIRE(n_nodes >= 1 && child_taxa >= 1, "scs_forest_split_level: need at least one node and one taxon");
    int32_t prev = 0;
    for (int32_t k = 0; k < n_nodes; ++k) {
        SCS_REQUIRE(node_tree_end[k] >= prev, "scs_forest_split_level: node_tree_end must not decrease");
        prev = node_tree_end[k];
    }
    SCS_REQUIRE(prev == f->n_trees, "scs_forest_split_level: the nodes must cover the forest's trees");
    level_args lv;
    lv.n_nodes = n_nodes;
    lv.node_tree_end = node_tree_end;
    lv.child_taxa = child_taxa;
    lv.child_trees = child_trees;
    lv.child_leaves = child_leaves;
    lv.present = present;
    lv.comp_root = comp_root;
    lv.sig = sig;
    return forest_split(ctx, f, part_of, new_id, n_parts, nullptr, strategy, out_union, info, false, &lv);
}

static int forest_split(scs_ctx *ctx, const scs_forest *f, const int32_t *part_of, const int32_t *new_id,
                        int32_t n_parts, const int32_t *part_taxa, int32_t strategy, scs_forest **out_forests,
                        scs_forest_info *info, bool force_serial, const level_args *lv) {
    SCS_REQUIRE(ctx && f && part_of && new_id && out_forests && info, "scs_forest_split: null argument");
    SCS_REQUIRE(n_parts >= 1 && n_parts <= SPLIT_MAX_PARTS, "scs_forest_split: 1 .. %d parts (asked: %d)",
                SPLIT_MAX_PARTS, n_parts);
    SCS_REQUIRE(strategy >= 0 && strategy <= 3, "scs_forest_split: strategy must be 0 .. 3");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int M = f->n_trees;
    const int64_t N = f->n_nodes, L = std::max<int64_t>(f->n_leaves, 1);
    const int32_t T = f->n_taxa;
    const bool level = lv != nullptr;
    const int32_t K = level ? lv->n_nodes : 0;
    // the children's taxon ids: per part 0 .. part_taxa[b] - 1; level mode: one universe for all parts
    const int32_t TC = level ? lv->child_taxa : T;
    if (level) {
        out_forests[0] = nullptr;
        // a level's forests stay resident until the walk has verified them; with the device nearly full the
        // runtime cannot even place a launch's private scratch (it aborts the process): refuse early, the caller
        // takes the subtree node by node (spectralclustersupertree_amd/levels.py)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + scs_arena_free_bytes(ctx->device) < ((size_t)6 << 30)) {
            scs_set_error("scs_forest_split_level: %.1f GB of device memory left", (double)free_b / (1u << 30));
            return SCS_ENOMEM;
        }
    } else {
        for (int b = 0; b < n_parts; ++b) {
            out_forests[b] = nullptr;
            SCS_REQUIRE(part_taxa[b] >= 0 && part_taxa[b] <= T, "scs_forest_split: bad taxon count of part %d", b);
        }
    }

    // scratch of this call (returned to the context's block cache at the end)
    struct scratch_t {
        scs_ctx *ctx;
        std::vector<void *> blocks;
        ~scratch_t() {
            for (void *p : blocks) scs_block_release(ctx, p);
        }
        int alloc(size_t bytes, void **out) {
            SCS_TRY(scs_block_alloc(ctx, bytes, out));
            blocks.push_back(*out);
            return SCS_OK;
        }
    } scratch{ctx, {}};
    auto region = std::make_shared<forest_region>();
    region->ctx = ctx;

    split_params p;
    memset(&p, 0, sizeof(p));
    p.n_trees = M;
    p.n_parts = n_parts;
    p.strategy = strategy;
    p.node_off = f->node_off;
    p.parent = f->parent;
    p.taxon = f->taxon;
    p.length = f->length;
    p.support = f->support;
    p.weights = f->weights;
    // ---- inputs: [flags (16 ints) | part_of | new_id | level mode: node_tree_end] in ONE copy
    const size_t in_ints = 16 + 2 * (size_t)T + (size_t)K;
    int32_t *d_in = nullptr;
    SCS_TRY(scratch.alloc(in_ints * 4, (void **)&d_in));
    {
        std::vector<int32_t> h_in(in_ints, 0);
        for (int b = 0; b < SPLIT_MAX_PARTS; ++b) h_in[1 + b] = 1;  // monotone until a negative length is met
        memcpy(h_in.data() + 16, part_of, (size_t)T * 4);
        memcpy(h_in.data() + 16 + T, new_id, (size_t)T * 4);
        if (level) memcpy(h_in.data() + 16 + 2 * (size_t)T, lv->node_tree_end, (size_t)K * 4);
        // (a pageable source is staged by the runtime before the call returns)
        SCS_HIP_CHECK(hipMemcpyAsync(d_in, h_in.data(), in_ints * 4, hipMemcpyHostToDevice, s));
    }
    p.flags = d_in;
    p.part_of = d_in + 16;
    p.new_id = d_in + 16 + T;
    const size_t pm = (size_t)n_parts * M;
    {
        // leaves_cnt | nodes_cnt | tree_pos (int32) then node_start | leaf_start (int64): one block
        unsigned char *blk = nullptr;
        SCS_TRY(scratch.alloc(pm * (3 * 4 + 2 * 8) + 64, (void **)&blk));
        p.node_start = (int64_t *)blk;
        p.leaf_start = p.node_start + pm;
        p.leaves_cnt = (int32_t *)(p.leaf_start + pm);
        p.nodes_cnt = p.leaves_cnt + pm;
        p.tree_pos = p.nodes_cnt + pm;
    }
    // ---- the children's node arrays: together the parts of a tree hold its leaves once and at most
    // leaves - 1 LCAs between them (an inner node of the parent may be kept by several parts)
    const int64_t NC = std::max<int64_t>(2 * L, 2);
    SCS_TRY(scratch.alloc((size_t)NC * 4, (void **)&p.cdepth));
    SCS_TRY(scratch.alloc((size_t)NC * 8, (void **)&p.cval));
    if (lv) SCS_TRY(scratch.alloc((size_t)n_parts * (M + 1) * 8, (void **)&p.c_node_off));
    else SCS_TRY(region->alloc((size_t)n_parts * (M + 1) * 8, (void **)&p.c_node_off));
    SCS_TRY(region->alloc((size_t)NC * 4, (void **)&p.c_parent));
    SCS_TRY(region->alloc((size_t)NC * 4, (void **)&p.c_taxon));
    SCS_TRY(region->alloc((size_t)NC * 8, (void **)&p.c_length));
    SCS_TRY(region->alloc((size_t)NC * 8, (void **)&p.c_support));
    // ---- everything the host wants back in ONE block (one copy to a page-locked twin of it):
    //   totals [n_parts][4] i64 | tree_off [n_parts][M + 1] i64 | weights [n_parts M] f64 | adj_val [L] f64 |
    //   tree_index [n_parts M] i32 | leaf_taxon [L] i32 | adj_depth [L] i32 | flags copy [16] i32 | present [n_parts][T] u8
    const size_t o_tot = 0, o_toff = o_tot + (size_t)n_parts * 4 * 8, o_w = o_toff + (size_t)n_parts * (M + 1) * 8,
                 o_aval = o_w + pm * 8, o_tidx = o_aval + (size_t)L * 8, o_ltax = o_tidx + pm * 4,
                 o_adep = o_ltax + (size_t)L * 4, o_flags = o_adep + (size_t)L * 4, o_pres = o_flags + 64,
                 out_bytes = (o_pres + (size_t)n_parts * T + 15) / 16 * 16;
    unsigned char *d_out = nullptr;
    // level mode: the tables stay on the device; the host gets ONE small block --
    //   totals [n_parts][4] i64 | child_leaves [n_parts][K] i64 | sig [TC][2] u64 | flags copy [16] i32 |
    //   child_trees [n_parts][K] i32 | comp_root [TC] i32 | present [TC] u8
    const size_t s_tot = 0, s_cl = s_tot + (size_t)n_parts * 4 * 8, s_sig = s_cl + (size_t)n_parts * K * 8,
                 s_flags = s_sig + (size_t)TC * 16, s_ct = s_flags + 64, s_root = s_ct + (size_t)n_parts * K * 4,
                 s_pres = s_root + (size_t)TC * 4, small_bytes = (s_pres + (size_t)TC + 15) / 16 * 16;
    unsigned char *d_small = nullptr;
    int64_t *u_node_off = nullptr, *u_tree_off = nullptr, *d_leaf_total = nullptr;
    if (!level) {
        SCS_TRY(region->alloc(out_bytes, (void **)&d_out));
        p.totals = (int64_t *)(d_out + o_tot);
        p.c_tree_off = (int64_t *)(d_out + o_toff);
        p.c_weights = (double *)(d_out + o_w);
        p.c_adj_val = (double *)(d_out + o_aval);
        p.c_tree_index = (int32_t *)(d_out + o_tidx);
        p.c_leaf_taxon = (int32_t *)(d_out + o_ltax);
        p.c_adj_depth = (int32_t *)(d_out + o_adep);
        p.c_present = d_out + o_pres;
        p.present_ld = T;
        SCS_HIP_CHECK(hipMemsetAsync(p.c_present, 0, (size_t)n_parts * T, s));
    } else {
        // (a kept tree has two or more leaves: at most L / 2 trees in all)
        const size_t mt = (size_t)std::min<int64_t>((int64_t)pm, L / 2 + 1);
        SCS_TRY(region->alloc(small_bytes + 64, (void **)&d_small));
        SCS_TRY(scratch.alloc((size_t)n_parts * (M + 1) * 8, (void **)&p.c_tree_off));
        SCS_TRY(region->alloc((mt + 1) * 8, (void **)&u_node_off));
        SCS_TRY(region->alloc((mt + 1) * 8, (void **)&u_tree_off));
        SCS_TRY(region->alloc(mt * 8, (void **)&p.c_weights));
        SCS_TRY(region->alloc(mt * 4, (void **)&p.c_tree_index));
        SCS_TRY(region->alloc((size_t)L * 8, (void **)&p.c_adj_val));
        SCS_TRY(region->alloc((size_t)L * 4, (void **)&p.c_leaf_taxon));
        SCS_TRY(region->alloc((size_t)L * 4, (void **)&p.c_adj_depth));
        p.totals = (int64_t *)(d_small + s_tot);
        d_leaf_total = (int64_t *)(d_small + small_bytes);
        p.c_present = d_small + s_pres;
        p.present_ld = 0;  // ONE array for all parts: the children's ids are unique across the parts
        SCS_HIP_CHECK(hipMemsetAsync(d_small, 0, small_bytes + 64, s));
    }

    // the offsets of every (part, tree): one workgroup while the trees are few, multi-block scans for the
    // millions of trees of a level forest
    auto split_scan = [&]() -> int {
        if (M <= 32768 || N >= ((int64_t)1 << 31) - 8) {
            k_split_scan<<<1, 1024, 0, s>>>(p);
            return SCS_OK;
        }
        int32_t *sc = nullptr, *bs = nullptr;
        const int64_t row = (int64_t)M + 1;
        SCS_TRY(scratch.alloc((size_t)3 * n_parts * row * 4, (void **)&sc));
        SCS_TRY(scratch.alloc((size_t)((row + 4095) / 4096 + 1) * 4 * SPLIT_MAX_PARTS, (void **)&bs));
        int32_t *sc_keep = sc, *sc_nodes = sc + (int64_t)n_parts * row, *sc_leaves = sc + (int64_t)2 * n_parts * row;
        SCS_TRY((scan_exclusive<SCAN_SUM>(f_cnt_pos{p.leaves_cnt, M}, sc_keep, row, n_parts, M, 0, bs, s)));
        SCS_TRY((scan_exclusive<SCAN_SUM>(f_cnt{p.nodes_cnt, M}, sc_nodes, row, n_parts, M, 0, bs, s)));
        SCS_TRY((scan_exclusive<SCAN_SUM>(f_cnt{p.leaves_cnt, M}, sc_leaves, row, n_parts, M, 0, bs, s)));
        k_split_finalize<<<dim3((unsigned)((M + 255) / 256), (unsigned)n_parts), 256, 0, s>>>(p, sc_keep, sc_nodes, sc_leaves);
        return SCS_OK;
    };
    // Big trees: every step per NODE (scans, walks); small ones: a thread per tree on an LDS copy.
    const int par_min = scs_dbg("SCS_FOREST_PARALLEL_MIN_TREE_NODES") ? atoi(scs_dbg("SCS_FOREST_PARALLEL_MIN_TREE_NODES")) : 32;
    const bool parallel = !force_serial && N < ((int64_t)1 << 31) - 8 && (double)N / M > (double)par_min;
    int32_t *c_tree_id = nullptr;
    if (parallel) {
        // (the union of a level makes its own tree ids on first use: c_tree_id numbers a part's trees from 0)
        if (level) SCS_TRY(scratch.alloc((size_t)NC * 4, (void **)&c_tree_id));
        else SCS_TRY(region->alloc((size_t)NC * 4, (void **)&c_tree_id));
        par_params q;
        memset(&q, 0, sizeof(q));
        q.n_trees = M;
        q.n_parts = n_parts;
        q.strategy = strategy;
        q.n_taxa = T;
        q.n_nodes = N;
        q.node_off = f->node_off;
        q.parent = f->parent;
        q.taxon = f->taxon;
        q.length = f->length;
        q.support = f->support;
        q.weights = f->weights;
        q.part_of = p.part_of;
        q.new_id = p.new_id;
        q.flags = p.flags;
        const unsigned gn = (unsigned)((N + 255) / 256), gm = (unsigned)((M + 255) / 256);
        if (!f->tree_id) {
            SCS_TRY(f->region->alloc((size_t)N * 4, (void **)&f->tree_id));
            k_par_tree_id<<<gn, 256, 0, s>>>(f->node_off, M, N, f->tree_id);
        }
        q.tree_id = f->tree_id;
        SCS_TRY(scratch.alloc((size_t)N, (void **)&q.pc));
        SCS_TRY(scratch.alloc((size_t)n_parts * (N + 1) * 4, (void **)&q.prev));
        SCS_TRY(scratch.alloc((size_t)n_parts * (N + 1) * 4, (void **)&q.rank));
        SCS_TRY(scratch.alloc((size_t)n_parts * (N + 1) * 4, (void **)&q.kpos));
        SCS_TRY(scratch.alloc((size_t)n_parts * N, (void **)&q.mark));
        SCS_TRY(scratch.alloc((size_t)n_parts * M, (void **)&q.keep));
        SCS_TRY(scratch.alloc((size_t)NC * 4, (void **)&q.corig));
        int32_t *block_sums = nullptr;
        SCS_TRY(scratch.alloc((size_t)((N + 1 + 4095) / 4096 + 1) * 4 * SPLIT_MAX_PARTS, (void **)&block_sums));
        q.cdepth = p.cdepth;
        q.cval = p.cval;
        q.leaves_cnt = p.leaves_cnt;
        q.nodes_cnt = p.nodes_cnt;
        q.tree_pos = p.tree_pos;
        q.node_start = p.node_start;
        q.leaf_start = p.leaf_start;
        q.totals = p.totals;
        q.c_node_off = p.c_node_off;
        q.c_tree_off = p.c_tree_off;
        q.c_parent = p.c_parent;
        q.c_taxon = p.c_taxon;
        q.c_tree_index = p.c_tree_index;
        q.c_tree_id = c_tree_id;
        q.c_length = p.c_length;
        q.c_support = p.c_support;
        q.c_weights = p.c_weights;
        q.c_leaf_taxon = p.c_leaf_taxon;
        q.c_adj_depth = p.c_adj_depth;
        q.c_adj_val = p.c_adj_val;
        q.c_present = p.c_present;
        q.present_ld = p.present_ld;
        k_par_pc<<<gn, 256, 0, s>>>(q);
        SCS_TRY((scan_exclusive<SCAN_MAX>(f_key{q.pc, 0}, q.prev, N + 1, n_parts, N, -1, block_sums, s)));
        SCS_TRY((scan_exclusive<SCAN_SUM>(f_ind{q.pc, 0}, q.rank, N + 1, n_parts, N, 0, block_sums, s)));
        k_par_keep<<<gm, 256, 0, s>>>(q);
        SCS_TRY((scan_exclusive<SCAN_SUM>(f_ind_kept{q.pc, q.keep, q.tree_id, M, 0}, q.rank, N + 1, n_parts, N, 0,
                                          block_sums, s)));
        SCS_HIP_CHECK(hipMemsetAsync(q.mark, 0, (size_t)n_parts * N, s));
        k_par_mark<<<gn, 256, 0, s>>>(q);
        SCS_TRY((scan_exclusive<SCAN_SUM>(f_mark{q.mark, N}, q.kpos, N + 1, n_parts, N, 0, block_sums, s)));
        k_par_nodes<<<gm, 256, 0, s>>>(q);
        p.tpb = SPLIT_THREADS;
        SCS_TRY(split_scan());
        k_par_nodes_fill<<<gn, 256, 0, s>>>(q);
        k_par_values<<<gn, 256, 0, s>>>(q);
        k_par_tables<<<gn, 256, 0, s>>>(q);
        k_par_trees<<<gm, 256, 0, s>>>(q);
        SCS_HIP_CHECK(hipGetLastError());
    } else {
        // scratch the size of the parent (only workgroups whose trees do not fit the LDS use it)
        SCS_TRY(scratch.alloc((size_t)N * 4, (void **)&p.stk_a));
        SCS_TRY(scratch.alloc((size_t)N * 4, (void **)&p.stk_b));
        SCS_TRY(scratch.alloc((size_t)N * 4, (void **)&p.stk_c));
        SCS_TRY(scratch.alloc((size_t)N, (void **)&p.mark));
        SCS_HIP_CHECK(hipMemsetAsync(p.mark, 0, (size_t)N, s));
        // trees per workgroup: as many as keep the workgroup's nodes inside the LDS copy
        int tpb = SPLIT_THREADS;
        while (tpb > 8 && (double)N / M * tpb > 0.85 * SPLIT_CAP) tpb >>= 1;
        p.tpb = tpb;
        const unsigned grid = (unsigned)((M + tpb - 1) / tpb);
        // (once per process: a recursion makes tens of thousands of splits; a part with less LDS per workgroup than
        // the staging area needs refuses here, and the caller restricts on the host -- ADVICE r05)
        SCS_REQUIRE((size_t)ctx->max_lds_bytes >= SPLIT_FILL_LDS,
                    "scs_forest_split: the device offers %d bytes of LDS per workgroup, the staged split needs %zu",
                    ctx->max_lds_bytes, SPLIT_FILL_LDS);
        static bool fill_attr_set = false;
        if (!fill_attr_set) {
            SCS_HIP_CHECK(hipFuncSetAttribute((const void *)k_split_fill, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)SPLIT_FILL_LDS));
            fill_attr_set = true;
        }
        k_split_count<<<grid, SPLIT_THREADS, 0, s>>>(p);
        SCS_TRY(split_scan());
        k_split_fill<<<grid, SPLIT_THREADS, SPLIT_FILL_LDS, s>>>(p);
        SCS_HIP_CHECK(hipGetLastError());
    }
    if (level) {
        // ---- the union of the children, what the host needs of it, and its analysis -- one small copy back
        int32_t *d_ct = (int32_t *)(d_small + s_ct);
        int64_t *d_cl = (int64_t *)(d_small + s_cl);
        k_level_union<<<dim3((unsigned)((M + 255) / 256), (unsigned)n_parts), 256, 0, s>>>(p, u_node_off, u_tree_off);
        k_level_counts<<<(unsigned)(((int64_t)n_parts * K + 255) / 256), 256, 0, s>>>(p, K, d_in + 16 + 2 * (size_t)T,
                                                                                     d_ct, d_cl);
        k_leaf_total<<<1, 64, 0, s>>>(p.totals, n_parts, d_leaf_total);
        SCS_HIP_CHECK(hipGetLastError());
        auto sc_alloc = [&](size_t bytes, void **out) { return scratch.alloc(bytes, out); };
        SCS_TRY(analyze_tables(sc_alloc, p.c_leaf_taxon, p.c_adj_depth, d_leaf_total, L, TC,
                               (int32_t *)(d_small + s_root), (unsigned long long *)(d_small + s_sig), s,
                               ctx->max_lds_bytes));
        SCS_HIP_CHECK(hipMemcpyAsync(d_small + s_flags, p.flags, 64, hipMemcpyDeviceToDevice, s));
        unsigned char *h_small = nullptr;
        SCS_TRY(scs_pinned_get(ctx, small_bytes, (void **)&h_small));
        struct pinned_guard {
            scs_ctx *ctx;
            void *p;
            ~pinned_guard() { scs_pinned_release(ctx, p); }
        } guard{ctx, h_small};
        SCS_HIP_CHECK(hipMemcpyAsync(h_small, d_small, small_bytes, hipMemcpyDeviceToHost, s));
        SCS_HIP_CHECK(hipStreamSynchronize(s));
        const int64_t *h_tot = (const int64_t *)(h_small + s_tot);
        const int32_t *h_flags = (const int32_t *)(h_small + s_flags);
        if (parallel && h_flags[9]) {
            // a root path longer than PAR_PATH (a comb): the one-thread-per-tree kernels take this split
            region.reset();
            return forest_split(ctx, f, part_of, new_id, n_parts, part_taxa, strategy, out_forests, info, true, lv);
        }
        if (h_flags[0] == -3) {
            scs_set_error("scs_forest_split: an internal node without support under the bootstrap weighting");
            return SCS_EUNSUP;
        }
        if (h_flags[0] != 0) {
            scs_set_error("scs_forest_split: malformed tree arrays (not preorder, or inconsistent leaf counts)");
            return SCS_EINVAL;
        }
        int64_t trees = 0, nodes = 0, leaves = 0;
        int32_t mono = 1;
        for (int b = 0; b < n_parts; ++b) {
            trees += h_tot[b * 4 + 0];
            nodes += h_tot[b * 4 + 1];
            leaves += h_tot[b * 4 + 2];
            mono = mono && h_flags[1 + b];
        }
        memcpy(lv->child_trees, h_small + s_ct, (size_t)n_parts * K * 4);
        memcpy(lv->child_leaves, h_small + s_cl, (size_t)n_parts * K * 8);
        memcpy(lv->present, h_small + s_pres, (size_t)TC);
        memcpy(lv->comp_root, h_small + s_root, (size_t)TC * 4);
        memcpy(lv->sig, h_small + s_sig, (size_t)TC * 16);
        auto c = std::unique_ptr<scs_forest>(new scs_forest());
        c->n_taxa = TC;
        c->n_trees = (int32_t)trees;
        c->n_nodes = nodes;
        c->n_leaves = leaves;
        c->region = region;
        c->node_off = u_node_off;
        c->tree_off = u_tree_off;
        c->parent = p.c_parent;
        c->taxon = p.c_taxon;
        c->length = p.c_length;
        c->support = p.c_support;
        c->weights = p.c_weights;
        c->tree_index = p.c_tree_index;
        c->leaf_taxon = p.c_leaf_taxon;
        c->adj_depth = p.c_adj_depth;
        c->adj_val = p.c_adj_val;
        c->present = p.c_present;
        c->has_tables = true;  // (on the device only: h_* stay null)
        c->tree_id = nullptr;  // (c_tree_id numbers a part's trees from 0: the union makes its own on first use)
        info[0].n_trees = (int32_t)trees;
        info[0].monotone = mono;
        info[0].n_nodes = nodes;
        info[0].n_leaves = leaves;
        out_forests[0] = c.release();
        return SCS_OK;
    }
    SCS_HIP_CHECK(hipMemcpyAsync(d_out + o_flags, p.flags, 64, hipMemcpyDeviceToDevice, s));
    unsigned char *h_out = nullptr;
    SCS_TRY(scs_pinned_get(ctx, out_bytes, (void **)&h_out));
    region->host_block = h_out;
    SCS_HIP_CHECK(hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    const int64_t *h_tot = (const int64_t *)(h_out + o_tot);
    const int32_t *h_flags = (const int32_t *)(h_out + o_flags);
    if (parallel && h_flags[9]) {
        // a root path longer than PAR_PATH (a comb): the one-thread-per-tree kernels take this split
        region.reset();
        return forest_split(ctx, f, part_of, new_id, n_parts, part_taxa, strategy, out_forests, info, true);
    }
    if (h_flags[0] == -3) {
        scs_set_error("scs_forest_split: an internal node without support under the bootstrap weighting");
        return SCS_EUNSUP;
    }
    if (h_flags[0] != 0) {
        scs_set_error("scs_forest_split: malformed tree arrays (not preorder, or inconsistent leaf counts)");
        return SCS_EINVAL;
    }
    int64_t tree_base = 0, leaf_base = 0;
    for (int b = 0; b < n_parts; ++b) {
        const int64_t trees = h_tot[b * 4 + 0], nodes = h_tot[b * 4 + 1], leaves = h_tot[b * 4 + 2],
                      node_base = h_tot[b * 4 + 3];
        auto c = std::unique_ptr<scs_forest>(new scs_forest());
        c->n_taxa = part_taxa[b];
        c->n_trees = (int32_t)trees;
        c->n_nodes = nodes;
        c->n_leaves = leaves;
        c->region = region;
        c->node_off = p.c_node_off + (int64_t)b * (M + 1);
        c->tree_off = p.c_tree_off + (int64_t)b * (M + 1);
        c->parent = p.c_parent + node_base;
        c->taxon = p.c_taxon + node_base;
        c->length = p.c_length + node_base;
        c->support = p.c_support + node_base;
        c->weights = p.c_weights + tree_base;
        c->tree_index = p.c_tree_index + tree_base;
        c->leaf_taxon = p.c_leaf_taxon + leaf_base;
        c->adj_depth = p.c_adj_depth + leaf_base;
        c->adj_val = p.c_adj_val + leaf_base;
        c->present = p.c_present + (int64_t)b * p.present_ld;
        c->has_tables = true;
        c->tree_id = parallel ? c_tree_id + node_base : nullptr;
        c->h_tree_off = (const int64_t *)(h_out + o_toff) + (int64_t)b * (M + 1);
        c->h_weights = (const double *)(h_out + o_w) + tree_base;
        c->h_tree_index = (const int32_t *)(h_out + o_tidx) + tree_base;
        c->h_leaf_taxon = (const int32_t *)(h_out + o_ltax) + leaf_base;
        c->h_adj_depth = (const int32_t *)(h_out + o_adep) + leaf_base;
        c->h_adj_val = (const double *)(h_out + o_aval) + leaf_base;
        c->h_present = h_out + o_pres + (int64_t)b * T;
        info[b].n_trees = (int32_t)trees;
        info[b].monotone = h_flags[1 + b];
        info[b].n_nodes = nodes;
        info[b].n_leaves = leaves;
        out_forests[b] = c.release();
        tree_base += trees;
        leaf_base += leaves;
    }
    return SCS_OK;
}

extern "C" int scs_forest_tables_host(scs_ctx *ctx, const scs_forest *f, const int64_t **tree_off,
                                      const int32_t **leaf_taxon, const int32_t **adj_depth,
                                      const double **adj_val, const int32_t **tree_index,
                                      const double **tree_w, const uint8_t **present) {
    SCS_REQUIRE(ctx && f && tree_off && leaf_taxon && adj_depth && adj_val && tree_index && tree_w && present,
                "scs_forest_tables_host: null argument");
    SCS_REQUIRE(f->has_tables && f->h_tree_off, "scs_forest_tables_host: only the children of scs_forest_split carry tables");
    *tree_off = f->h_tree_off;
    *leaf_taxon = f->h_leaf_taxon;
    *adj_depth = f->h_adj_depth;
    *adj_val = f->h_adj_val;
    *tree_index = f->h_tree_index;
    *tree_w = f->h_weights;
    *present = f->h_present;
    return SCS_OK;
}

extern "C" int scs_forest_tables_download(scs_ctx *ctx, const scs_forest *f, int64_t *tree_off, int32_t *leaf_taxon,
                                          int32_t *adj_depth, double *adj_val, int32_t *tree_index,
                                          double *tree_w, uint8_t *present) {
    SCS_REQUIRE(ctx && f, "scs_forest_tables_download: null argument");
    SCS_REQUIRE(f->has_tables, "scs_forest_tables_download: only the children of scs_forest_split carry tables");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t m = (size_t)f->n_trees, l = (size_t)f->n_leaves;
    if (tree_off) SCS_HIP_CHECK(hipMemcpyAsync(tree_off, f->tree_off, (m + 1) * 8, hipMemcpyDeviceToHost, s));
    if (leaf_taxon && l) SCS_HIP_CHECK(hipMemcpyAsync(leaf_taxon, f->leaf_taxon, l * 4, hipMemcpyDeviceToHost, s));
    if (adj_depth && l) SCS_HIP_CHECK(hipMemcpyAsync(adj_depth, f->adj_depth, l * 4, hipMemcpyDeviceToHost, s));
    if (adj_val && l) SCS_HIP_CHECK(hipMemcpyAsync(adj_val, f->adj_val, l * 8, hipMemcpyDeviceToHost, s));
    if (tree_index && m) SCS_HIP_CHECK(hipMemcpyAsync(tree_index, f->tree_index, m * 4, hipMemcpyDeviceToHost, s));
    if (tree_w && m) SCS_HIP_CHECK(hipMemcpyAsync(tree_w, f->weights, m * 8, hipMemcpyDeviceToHost, s));
    if (present && f->n_taxa) SCS_HIP_CHECK(hipMemcpyAsync(present, f->present, (size_t)f->n_taxa, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    return SCS_OK;
}

extern "C" int scs_forest_download(scs_ctx *ctx, const scs_forest *f, int32_t t_begin, int32_t t_end,
                                   int64_t *node_off, int32_t *parent, int32_t *taxon, double *length,
                                   double *support, double *weights) {
    SCS_REQUIRE(ctx && f && node_off, "scs_forest_download: null argument");
    SCS_REQUIRE(t_begin >= 0 && t_begin <= t_end && t_end <= f->n_trees, "scs_forest_download: bad tree range");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t m = (size_t)(t_end - t_begin);
    SCS_HIP_CHECK(hipMemcpyAsync(node_off, f->node_off + t_begin, (m + 1) * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    const int64_t lo = node_off[0], n = node_off[m] - lo;
    if (parent && n) SCS_HIP_CHECK(hipMemcpyAsync(parent, f->parent + lo, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (taxon && n) SCS_HIP_CHECK(hipMemcpyAsync(taxon, f->taxon + lo, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (length && n) SCS_HIP_CHECK(hipMemcpyAsync(length, f->length + lo, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (support && n) SCS_HIP_CHECK(hipMemcpyAsync(support, f->support + lo, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    if (weights && m) SCS_HIP_CHECK(hipMemcpyAsync(weights, f->weights + t_begin, m * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t t = 0; t <= m; ++t) node_off[t] -= lo;  // (relative to the first downloaded tree)
    return SCS_OK;
}


// Components and contraction signatures of a forest that carries tables (a child of scs_forest_split or the
// union of scs_forest_split_level), over its n_taxa ids -- what scs_forest_split_level reports for its
// children, for a level's first forest (reference: scs.py:122, :302-316).
extern "C" int scs_forest_analyze(scs_ctx *ctx, const scs_forest *f, int32_t *comp_root, uint64_t *sig) {
    SCS_REQUIRE(ctx && f && comp_root && sig, "scs_forest_analyze: null argument");
    SCS_REQUIRE(f->has_tables, "scs_forest_analyze: the forest carries no tables (not a child of scs_forest_split)");
    SCS_REQUIRE(f->n_leaves < ((int64_t)1 << 31) - 8, "scs_forest_analyze: too many leaves");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    struct scratch_t {
        scs_ctx *ctx;
        std::vector<void *> blocks;
        ~scratch_t() {
            for (void *p : blocks) scs_block_release(ctx, p);
        }
        int alloc(size_t bytes, void **out) {
            SCS_TRY(scs_block_alloc(ctx, bytes, out));
            blocks.push_back(*out);
            return SCS_OK;
        }
    } scratch{ctx, {}};
    const int32_t T = f->n_taxa;
    const size_t o_sig = 0, o_root = (size_t)T * 16, o_n = (o_root + (size_t)T * 4 + 15) / 16 * 16, bytes = o_n + 16;
    unsigned char *d = nullptr;
    SCS_TRY(scratch.alloc(bytes, (void **)&d));
    const int64_t L = f->n_leaves;
    SCS_HIP_CHECK(hipMemcpyAsync(d + o_n, &L, 8, hipMemcpyHostToDevice, s));
    auto sc_alloc = [&](size_t b, void **out) { return scratch.alloc(b, out); };
    SCS_TRY(analyze_tables(sc_alloc, f->leaf_taxon, f->adj_depth, (const int64_t *)(d + o_n), L, T,
                           (int32_t *)(d + o_root), (unsigned long long *)(d + o_sig), s, ctx->max_lds_bytes));
    SCS_HIP_CHECK(hipMemcpyAsync(sig, d + o_sig, (size_t)T * 16, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipMemcpyAsync(comp_root, d + o_root, (size_t)T * 4, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    return SCS_OK;
}

// tables of the trees [t_begin, t_end) of a forest that carries them (tree_off made relative to the first of
// them; any output may be null): the exact host routines of a flagged node (contraction groups)
extern "C" int scs_forest_tables_download_range(scs_ctx *ctx, const scs_forest *f, int32_t t_begin, int32_t t_end,
                                                int64_t *tree_off, int32_t *leaf_taxon, int32_t *adj_depth,
                                                double *adj_val, double *tree_w) {
    SCS_REQUIRE(ctx && f && tree_off, "scs_forest_tables_download_range: null argument");
    SCS_REQUIRE(f->has_tables, "scs_forest_tables_download_range: the forest carries no tables");
    SCS_REQUIRE(t_begin >= 0 && t_begin <= t_end && t_end <= f->n_trees, "scs_forest_tables_download_range: bad tree range");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t m = (size_t)(t_end - t_begin);
    SCS_HIP_CHECK(hipMemcpyAsync(tree_off, f->tree_off + t_begin, (m + 1) * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    const int64_t lo = tree_off[0], l = tree_off[m] - lo;
    if (leaf_taxon && l) SCS_HIP_CHECK(hipMemcpyAsync(leaf_taxon, f->leaf_taxon + lo, (size_t)l * 4, hipMemcpyDeviceToHost, s));
    if (adj_depth && l) SCS_HIP_CHECK(hipMemcpyAsync(adj_depth, f->adj_depth + lo, (size_t)l * 4, hipMemcpyDeviceToHost, s));
    if (adj_val && l) SCS_HIP_CHECK(hipMemcpyAsync(adj_val, f->adj_val + lo, (size_t)l * 8, hipMemcpyDeviceToHost, s));
    if (tree_w && m) SCS_HIP_CHECK(hipMemcpyAsync(tree_w, f->weights + t_begin, m * 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    for (size_t t = 0; t <= m; ++t) tree_off[t] -= lo;
    return SCS_OK;
}


__global__ void k_slice_offsets(const int64_t *__restrict__ src, int64_t n, int64_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] - src[0];
}

// The trees [t_begin, t_end) of a forest as a forest of its own WITHOUT copying a node: the slice points into
// the arrays of `f` (and keeps them alive), only the offsets are made anew.  Taxon ids stay those of `f`.
// One node of a level forest whose subtree the recursion takes up again from its own trees (a provisional
// partition that the true draws did not confirm, levels.py).
extern "C" int scs_forest_slice(scs_ctx *ctx, const scs_forest *f, int32_t t_begin, int32_t t_end, scs_forest **out) {
    SCS_REQUIRE(ctx && f && out, "scs_forest_slice: null argument");
    SCS_REQUIRE(t_begin >= 0 && t_begin < t_end && t_end <= f->n_trees, "scs_forest_slice: bad tree range");
    SCS_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int32_t m = t_end - t_begin;
    auto c = std::unique_ptr<scs_forest>(new scs_forest());
    c->region = std::make_shared<forest_region>();
    c->region->ctx = ctx;
    c->region->base = f->region;
    int64_t ends[4] = {0, 0, 0, 0};  // node_off[t_begin], node_off[t_end], tree_off[t_begin], tree_off[t_end]
    SCS_HIP_CHECK(hipMemcpyAsync(&ends[0], f->node_off + t_begin, 8, hipMemcpyDeviceToHost, s));
    SCS_HIP_CHECK(hipMemcpyAsync(&ends[1], f->node_off + t_end, 8, hipMemcpyDeviceToHost, s));
    if (f->has_tables) {
        SCS_HIP_CHECK(hipMemcpyAsync(&ends[2], f->tree_off + t_begin, 8, hipMemcpyDeviceToHost, s));
        SCS_HIP_CHECK(hipMemcpyAsync(&ends[3], f->tree_off + t_end, 8, hipMemcpyDeviceToHost, s));
    }
    SCS_TRY(c->region->alloc((size_t)(m + 1) * 8, (void **)&c->node_off));
    k_slice_offsets<<<(unsigned)((m + 1 + 255) / 256), 256, 0, s>>>(f->node_off + t_begin, (int64_t)m + 1, c->node_off);
    if (f->has_tables) {
        SCS_TRY(c->region->alloc((size_t)(m + 1) * 8, (void **)&c->tree_off));
        k_slice_offsets<<<(unsigned)((m + 1 + 255) / 256), 256, 0, s>>>(f->tree_off + t_begin, (int64_t)m + 1, c->tree_off);
    }
    SCS_HIP_CHECK(hipGetLastError());
    SCS_HIP_CHECK(hipStreamSynchronize(s));
    const int64_t n0 = ends[0], l0 = ends[2];
    c->n_taxa = f->n_taxa;
    c->n_trees = m;
    c->n_nodes = ends[1] - ends[0];
    c->n_leaves = f->has_tables ? ends[3] - ends[2] : 0;
    c->parent = f->parent + n0;
    c->taxon = f->taxon + n0;
    c->length = f->length + n0;
    c->support = f->support + n0;
    c->weights = f->weights + t_begin;
    c->has_tables = f->has_tables;
    if (f->has_tables) {
        c->leaf_taxon = f->leaf_taxon + l0;
        c->adj_depth = f->adj_depth + l0;
        c->adj_val = f->adj_val + l0;
        c->tree_index = f->tree_index ? f->tree_index + t_begin : nullptr;
        c->present = f->present;
    } else {
        // (the leaf count sizes a split's outputs: without tables it is not known here -- count the leaves)
        c->n_leaves = c->n_nodes;
    }
    *out = c.release();
    return SCS_OK;
}
