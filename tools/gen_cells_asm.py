#!/usr/bin/env python3
"""Generates spectralclustersupertree_amd/csrc/scs_cells_asm.h: the hand-scheduled 64-cell
loops of the tile kernels, each as ONE inline-asm statement.

SCS_CELLS_ASM (k_accumulate_mono).  Per cell: ds_read_b64 (row nb of the row-row table,
contiguous per lane), v_min_f64 with the column's own LCA value, v_add_f64 into the cell's
accumulator -- in tree order, unfused.

SCS_CELLS_GEN_ASM (k_accumulate_gen, values that are not monotone in the depth).  Per cell: the
same read; the rows whose sorted rank lies in the column's interval [TLO, TLO + TW] take the
column's own LCA value instead -- the rank of row i is wave-uniform (s_bfe_u32 out of 16 packed
SGPRs), the lanes inside the interval are selected through EXEC (v_cmpx + v_mov_b64), then
v_add_f64.

DEPTH reads stay in flight (plain ds_read_b64: the compiler's ds_read2_b64 pairs run at half
the LDS rate and conflict 4-way on this access pattern), each waited for with a counted
lgkmcnt just before its use and re-issued right after.
"""
import sys

ROWS = 64
DEPTH = 8      # reads in flight, monotone loop
GEN_DEPTH = 5  # general loop: two fewer (its kernel is short of registers at 3 waves per SIMD)
BS = " \\\n"


def quoted(lines):
    return BS.join(f'        "{ln}\\n\\t"' for ln in lines)


def mono():
    lines = ["s_waitcnt lgkmcnt(0)"]
    for k in range(DEPTH):
        lines.append(f"ds_read_b64 %[t{k}], %[addr] offset:{8 * k}")
    for i in range(ROWS):
        k = i % DEPTH
        outstanding_after = min(DEPTH - 1, ROWS - 1 - i)
        lines.append(f"s_waitcnt lgkmcnt({outstanding_after})")
        lines.append(f"v_min_f64 %[t{k}], %[t{k}], %[vn]")
        lines.append(f"v_add_f64 %[a{i}], %[a{i}], %[t{k}]")
        if i + DEPTH < ROWS:
            lines.append(f"ds_read_b64 %[t{k}], %[addr] offset:{8 * (i + DEPTH)}")
    outs = ("," + BS).join(f'          [a{i}] "+v"(ACC[{i}])' for i in range(ROWS))
    tmps = ("," + BS).join(f'          [t{k}] "=&v"(TMP[{k}])' for k in range(DEPTH))
    return f"""#define SCS_CELLS_ASM(ACC, TMP, ADDR, VN)                                                   \\
    asm volatile(                                                                            \\
{quoted(lines)} \\
        :                                                                                    \\
{outs}, \\
{tmps} \\
        : [addr] "v"(ADDR), [vn] "v"(VN)                                                     \\
        : "memory")
"""


def general():
    lines = ["s_waitcnt lgkmcnt(0)"]
    for k in range(GEN_DEPTH):
        lines.append(f"ds_read_b64 %[t{k}], %[addr] offset:{8 * k}")
    for i in range(ROWS):
        k = i % GEN_DEPTH
        outstanding_after = min(GEN_DEPTH - 1, ROWS - 1 - i)
        # rank of row i: byte i & 3 of packed SGPR i >> 2
        lines.append(f"s_bfe_u32 %[sr], %[r{i >> 2}], {hex(((i & 3) * 8) | (8 << 16))}")
        lines.append(f"v_sub_u32 %[tt], %[sr], %[tlo]")
        lines.append(f"s_waitcnt lgkmcnt({outstanding_after})")
        lines.append(f"v_cmpx_ge_u32 vcc, %[tw], %[tt]")
        lines.append(f"v_mov_b64 %[t{k}], %[vv]")
        lines.append("s_mov_b64 exec, -1")
        lines.append(f"v_add_f64 %[a{i}], %[a{i}], %[t{k}]")
        if i + GEN_DEPTH < ROWS:
            lines.append(f"ds_read_b64 %[t{k}], %[addr] offset:{8 * (i + GEN_DEPTH)}")
    outs = ("," + BS).join(f'          [a{i}] "+v"(ACC[{i}])' for i in range(ROWS))
    tmps = ("," + BS).join(f'          [t{k}] "=&v"(TMP[{k}])' for k in range(GEN_DEPTH))
    ranks = ", ".join(f'[r{j}] "s"(RP[{j}])' for j in range(ROWS // 4))
    return f"""#define SCS_CELLS_GEN_ASM(ACC, TMP, TT, SR, ADDR, VV, TLO, TW, RP)                          \\
    asm volatile(                                                                            \\
{quoted(lines)} \\
        :                                                                                    \\
{outs}, \\
{tmps}, \\
          [tt] "=&v"(TT), [sr] "=&s"(SR)                                                     \\
        : [addr] "v"(ADDR), [vv] "v"(VV), [tlo] "v"(TLO), [tw] "v"(TW),                      \\
          {ranks} \\
        : "memory", "vcc")
"""


def cells_expand():
    """SCS_CELLS_EXPAND_ASM (k_accumulate_pipe): the 64 cells of tree t from one table buffer
    with the wave's eight expansion steps of tree t + 1's table (the other buffer) woven in, one
    step per eight cells.  A step (rank b = B0 + j): the gap value g[b] and the row of rank b
    come out of the lanes that hold them (v_readlane -> SGPRs), the lanes whose rank is <= b
    store their running minimum at (own row, row of b) and its mirror image, then take g[b] in.
    Hazards are this generator's to keep (the compiler's recognizer does not look inside
    inline asm; gfx940 family: a VALU-written SGPR needs 2 wait states before a VALU reads it,
    a VALU-written EXEC 4 before v_readlane): the v_readlane results are consumed a cell (four
    instructions) later, a step's v_cmpx is followed by seven cells before the next v_readlane.
    The counted lgkmcnt waits are exact: the generator keeps the queue of LDS operations
    (reads and stores return in order)."""
    lines = ["s_waitcnt lgkmcnt(0)"]
    queue = []  # outstanding LDS operations in issue order: ("r", k) reads into t{k}, ("w",)

    def read(k, row):
        lines.append(f"ds_read_b64 %[t{k}], %[addr] offset:{8 * row}")
        queue.append(("r", k))

    def consume(k):
        # wait until the read into t{k} is done: everything up to it has left the queue
        idx = next(i for i, op in enumerate(queue) if op == ("r", k))
        younger = len(queue) - 1 - idx
        lines.append(f"s_waitcnt lgkmcnt({younger})")
        del queue[: idx + 1]

    def store(addr_reg):
        lines.append(f"ds_write_b64 {addr_reg}, %[cur]")
        queue.append(("w",))

    for k in range(DEPTH):
        read(k, k)
    for i in range(ROWS):
        k = i % DEPTH
        j, ph = divmod(i, 8)
        if ph == 0:
            # rank b of this step, the gap value and the row of rank b (SGPRs 92, 94:95, 96)
            lines.append(f"s_add_u32 s92, %[b0], {j}")
            lines.append("v_readlane_b32 s94, %[glo], s92")
            lines.append("v_readlane_b32 s95, %[ghi], s92")
            lines.append("v_readlane_b32 s96, %[so], s92")
        consume(k)
        lines.append(f"v_min_f64 %[t{k}], %[t{k}], %[vn]")
        lines.append(f"v_add_f64 %[a{i}], %[a{i}], %[t{k}]")
        if i + DEPTH < ROWS:
            read(k, i + DEPTH)
        if ph == 0:
            lines.append("s_mul_i32 s97, s96, %[ld8]")
            lines.append("v_lshl_add_u32 %[x1], s96, 3, %[rowb]")
            lines.append("v_add_u32 %[x2], s97, %[colb]")
            lines.append("v_cmpx_ge_u32 vcc, s92, %[rho]")
            store("%[x1]")
            store("%[x2]")
            lines.append("v_min_f64 %[cur], %[cur], s[94:95]")
            lines.append("s_mov_b64 exec, -1")
    outs = ("," + BS).join(f'          [a{i}] "+v"(ACC[{i}])' for i in range(ROWS))
    tmps = ("," + BS).join(f'          [t{k}] "=&v"(TMP[{k}])' for k in range(DEPTH))
    return f"""#define SCS_CELLS_EXPAND_ASM(ACC, TMP, X1, X2, ADDR, VN, CUR, GLO, GHI, SO, RHO, ROWB, COLB, B0, LD8) \\
    asm volatile( \\
{quoted(lines)} \\
        : \\
{outs}, \\
{tmps}, \\
          [x1] "=&v"(X1), [x2] "=&v"(X2), [cur] "+v"(CUR) \\
        : [addr] "v"(ADDR), [vn] "v"(VN), [glo] "v"(GLO), [ghi] "v"(GHI), [so] "v"(SO), \\
          [rho] "v"(RHO), [rowb] "v"(ROWB), [colb] "v"(COLB), [b0] "s"(B0), [ld8] "s"(LD8) \\
        : "memory", "vcc", "s92", "s94", "s95", "s96", "s97")
"""


def main(path):
    text = f"""// GENERATED by tools/gen_cells_asm.py -- do not edit.
// SCS_CELLS_ASM(ACC, TMP, ADDR, VN): ACC double[{ROWS}] accumulators, TMP double[{DEPTH}] scratch,
// ADDR the LDS byte address of the lane's table row, VN the column's own LCA value.
// SCS_CELLS_GEN_ASM(ACC, TMP, TT, SR, ADDR, VV, TLO, TW, RP): as above for values that are not
// monotone in the depth: rows whose sorted rank r has r - TLO <= TW (unsigned) take VV instead
// of the table entry; RP int[{ROWS // 4}] wave-uniform, byte i & 3 of RP[i >> 2] = rank of row i;
// TT (unsigned, vector) and SR (int, scalar) are scratch.  All 64 lanes must be active.
// SCS_CELLS_EXPAND_ASM(ACC, TMP, X1, X2, ADDR, VN, CUR, GLO, GHI, SO, RHO, ROWB, COLB, B0, LD8): the
// monotone cell loop with the wave's eight expansion steps of the NEXT tree's table woven in
// (k_accumulate_pipe): CUR (double, in/out) the lane's running minimum, GLO / GHI the halves of
// g[lane], SO the row of rank `lane`, RHO the lane's rank, ROWB / COLB the LDS byte addresses
// of the lane's row and column of the other table buffer, B0 (scalar) the wave's first rank,
// LD8 (scalar) the table's row stride in bytes; X1, X2 (unsigned, vector) are scratch.
#pragma once
#define SCS_CELLS_DEPTH {DEPTH}
#define SCS_CELLS_GEN_DEPTH {GEN_DEPTH}
{mono()}
{general()}
{cells_expand()}"""
    open(path, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "spectralclustersupertree_amd/csrc/scs_cells_asm.h")
