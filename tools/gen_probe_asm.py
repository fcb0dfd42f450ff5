#!/usr/bin/env python3
"""Generates tools/cells_probe_asm.h: the inner loop of a ROW-lane cell kernel for
tools/cells_probe.hip (a measurement, not product code).

PROBE_D_ASM: lanes = the 64 rows of a tile, one wave walks 64 columns.  Per column: the LDS
byte offset of the table row next to the column and the column's own LCA value are wave-uniform
and come in through the SCALAR path (s_load from a region this wave filled with vector stores
earlier: s_dcache_inv first), the table row is read conflict-free (64 consecutive doubles),
v_min_f64 with the scalar value, v_add_f64.  Eight groups of eight columns; the scalars of group
g + 1 (values) and g + 2 (offsets) are requested a group ahead; every group boundary is one
s_waitcnt lgkmcnt(0) (scalar loads return out of order: no counted wait may stand for them).
SGPRs are fixed: values s[36:51] / s[52:67], offsets s[68:75] / s[76:83] / s[84:91].
"""
import sys

BS = " \\\n"
VN = [36, 52]
OFF = [68, 76, 84]


def q(lines):
    return BS.join(f'        "{ln}\\n\\t"' for ln in lines)


def probe_d():
    L = ["s_dcache_inv",
         f"s_load_dwordx16 s[{VN[0]}:{VN[0] + 15}], %[base], 0x0",
         f"s_load_dwordx8 s[{OFF[0]}:{OFF[0] + 7}], %[base], 0x200",
         f"s_load_dwordx8 s[{OFF[1]}:{OFF[1] + 7}], %[base], 0x220",
         "s_waitcnt lgkmcnt(0)"]
    for j in range(8):
        L.append(f"v_add_u32 %[x{j}], s{OFF[0] + j}, %[lane8]")
        L.append(f"ds_read_b64 %[t{j}], %[x{j}]")
    L.append(f"s_load_dwordx16 s[{VN[1]}:{VN[1] + 15}], %[base], 0x40")
    L.append(f"s_load_dwordx8 s[{OFF[2]}:{OFF[2] + 7}], %[base], 0x240")
    for g in range(8):
        L.append("s_waitcnt lgkmcnt(0)")
        vn = VN[g & 1]
        nxt = OFF[(g + 1) % 3]
        for j in range(8):
            c = 8 * g + j
            L.append(f"v_min_f64 %[t{j}], %[t{j}], s[{vn + 2 * j}:{vn + 2 * j + 1}]")
            L.append(f"v_add_f64 %[a{c}], %[a{c}], %[t{j}]")
            if g < 7:
                L.append(f"v_add_u32 %[x{j}], s{nxt + j}, %[lane8]")
                L.append(f"ds_read_b64 %[t{j}], %[x{j}]")
        if 1 <= g <= 6:  # stand-in for the table expansion of the next tree: two stores per step
            L.append("ds_write_b64 %[w1], %[wv]")
            L.append("ds_write_b64 %[w2], %[wv]")
        if g + 2 < 8:
            L.append(f"s_load_dwordx16 s[{vn}:{vn + 15}], %[base], {hex(64 * (g + 2))}")
        if g + 3 < 8:
            o = OFF[g % 3]
            L.append(f"s_load_dwordx8 s[{o}:{o + 7}], %[base], {hex(0x200 + 32 * (g + 3))}")
    outs = ("," + BS).join(f'          [a{i}] "+v"(ACC[{i}])' for i in range(64))
    tmps = ("," + BS).join(f'          [t{k}] "=&v"(TMP[{k}]), [x{k}] "=&v"(ADR[{k}])' for k in range(8))
    clob = ", ".join(f'"s{r}"' for r in range(36, 92))
    return f"""#define PROBE_D_ASM(ACC, TMP, ADR, LANE8, BASE, W1, W2, WV) \\
    asm volatile( \\
{q(L)} \\
        : \\
{outs}, \\
{tmps} \\
        : [lane8] "v"(LANE8), [base] "s"(BASE), [w1] "v"(W1), [w2] "v"(W2), [wv] "v"(WV) \\
        : "memory", {clob})
"""




ACC0 = 64  # first accumulator register of PROBE_H_ASM: acc[i] = v[ACC0 + 2 i : ACC0 + 2 i + 1]


def probe_h(name="PROBE_H_ASM", depth=8, group=8, weave=True, nops=0, mode="onoff", same_idx=False, read2=False, noidx=0):
    """The rank-halved cell loop (round 5).  The statement walks the 64 sorted RANKS of
    the tile's rows; ranks 0-31 read the lane's row of half-table A (33 rows x 32 columns, row stride
    33 doubles: conflict-free), ranks 32-63 of half-table B; the cell is min(entry, cap of the half)
    and goes into the accumulator of the rank's ORIGINAL row, which is wave-uniform and changes per
    tree: VGPR index mode (SRC0 | DST), M0 written straight from 16-bit entries
    0x9000 | 2 orig(r) packed two per SGPR.  The mins of `group` cells run unindexed, the
    indexed adds follow (every VALU operand of the enabled kinds is indexed while the mode is
    on), each add re-issuing the read of the cell `depth` ranks on.
    Variants: `nops` wait states between an M0 write and the indexed add; mode "onoff" =
    s_set_gpr_idx_on / off around every group of adds, "once" = the enable bit set once per
    statement and M0 = 0 (no operand kind enabled) while the mins run; same_idx = one index per
    group (WRONG sums: prices the per-cell M0 write); read2 = one ds_read2_b64 per two cells."""
    L = ["s_waitcnt lgkmcnt(0)"]
    queue = []
    step = 2 if read2 else 1

    def read(k, r):
        half = "B" if r >= 32 else "A"
        if read2:
            o = r % 32
            L.append(f"ds_read2_b64 %[u{k // 2}], %[addr{half}] offset0:{o} offset1:{o + 1}")
        else:
            L.append(f"ds_read_b64 %[t{k}], %[addr{half}] offset:{8 * (r % 32)}")
        queue.append(("r", k))

    def consume(k):
        idx = next(i for i, op in enumerate(queue) if op == ("r", k))
        younger = len(queue) - 1 - idx
        L.append(f"s_waitcnt lgkmcnt({min(younger, 15)})")  # (the counter has four bits)
        del queue[: idx + 1]

    def entry(r):
        return f"%[p{r >> 1}], {hex(((r & 1) * 16) | (16 << 16))}"

    def nop():
        if nops:
            L.append(f"s_nop {nops - 1}")

    def tmp(k):
        if read2:
            return f"%[u{k // 2}]" if False else (f"%[lo{k // 2}]" if k % 2 == 0 else f"%[hi{k // 2}]")
        return f"%[t{k}]"

    for k in range(0, depth, step):
        read(k, k)
    if mode == "once" and not noidx:
        L.append("s_mov_b32 %[st], 0")
        L.append("s_set_gpr_idx_on %[st], 0x0")
        nop()
    for g in range(64 // group):
        r0 = g * group
        cap = "capB" if r0 >= 32 else "capA"
        for j in range(group):
            k = (r0 + j) % depth
            if not read2 or j % 2 == 0:
                consume(k)
            L.append(f"v_min_f64 {tmp(k)}, {tmp(k)}, %[{cap}]")
        if mode == "onoff":
            L.append(f"s_bfe_u32 %[st], {entry(r0)}")
            L.append("s_set_gpr_idx_on %[st], 0x9")
            nop()
        for j in range(group):
            r = r0 + j
            k = r % depth
            if (j or mode == "once") and not (same_idx and j) and noidx < 2:
                L.append(f"s_bfe_u32 m0, {entry(r)}")
                nop()
            a = ACC0 + 2 * r if noidx else ACC0
            L.append(f"v_add_f64 v[{a}:{a + 1}], v[{a}:{a + 1}], {tmp(k)}")
            if r + depth < 64 and (not read2 or j % 2 == 1):
                read(k - (1 if read2 else 0), r + depth - (1 if read2 else 0))
        if mode == "onoff":
            L.append("s_set_gpr_idx_off")
        elif noidx < 2:
            L.append("s_mov_b32 m0, 0")
        nop()
        if weave and g * group % 8 == 0 and 1 <= g * group // 8 <= 4:  # stand-in for the wave's four expansion steps of the next tree
            L.append("ds_write_b64 %[w1], %[wv]")
            queue.append(("w",))
            L.append("ds_write_b64 %[w2], %[wv]")
            queue.append(("w",))
    if mode == "once" and not noidx:
        L.append("s_set_gpr_idx_off")
    outs = ("," + BS).join(f'          "+{{v[{ACC0 + 2 * i}:{ACC0 + 2 * i + 1}]}}"(ACC[{i}])' for i in range(64))
    if read2:
        # a pair of temporaries is one 128-bit operand; its halves are named through register-tuple
        # subscripts, which inline asm cannot express -- so read2 variants bind the temporaries to
        # fixed registers below the accumulators
        base = ACC0 - 2 * depth
        text = BS.join(f'        "{ln}\\n\\t"' for ln in L)
        for k2 in range(depth // 2):
            text = text.replace(f"%[u{k2}]", f"v[{base + 4 * k2}:{base + 4 * k2 + 3}]")
            text = text.replace(f"%[lo{k2}]", f"v[{base + 4 * k2}:{base + 4 * k2 + 1}]")
            text = text.replace(f"%[hi{k2}]", f"v[{base + 4 * k2 + 2}:{base + 4 * k2 + 3}]")
        tmps = ("," + BS).join(f'          "=&{{v[{base + 2 * k}:{base + 2 * k + 1}]}}"(TMP[{k}])' for k in range(depth))
    else:
        text = q(L)
        tmps = ("," + BS).join(f'          [t{k}] "=&v"(TMP[{k}])' for k in range(depth))
    perm = ", ".join(f'[p{j}] "s"(PERM[{j}])' for j in range(32))
    return f"""#define {name}(ACC, TMP, ST, ADDRA, ADDRB, CAPA, CAPB, PERM, W1, W2, WV) \\
    asm volatile( \\
{text} \\
        : \\
{outs}, \\
{tmps}, \\
          [st] "=&s"(ST) \\
        : [addrA] "v"(ADDRA), [addrB] "v"(ADDRB), [capA] "v"(CAPA), [capB] "v"(CAPB), \\
          {perm}, \\
          [w1] "v"(W1), [w2] "v"(W2), [wv] "v"(WV) \\
        : "memory")
"""



def probe_l(name, b128=False, wait_every=1, depth=8, tbase=8, rows=64):
    """Variants of the product's column-lane cell loop (SCS_CELLS_ASM) with FEWER INSTRUCTIONS per
    cell -- round 5's probe found the tile kernels bound by instruction issue (about one instruction
    of any kind per four cycles and SIMD), not by the LDS: `wait_every` cells share one counted
    s_waitcnt; b128 = one ds_read_b128 per two cells (two neighbouring entries of the lane's table
    row: the row stride must be even, 66 doubles).  Temporaries are fixed registers v[tbase ...]
    (the halves of a 128-bit destination have to be named)."""
    L = ["s_waitcnt lgkmcnt(0)"]
    queue = []
    per = 2 if b128 else 1

    def t(k):
        return f"v[{tbase + 2 * k}:{tbase + 2 * k + 1}]"

    def read(k, row):
        if b128:
            L.append(f"ds_read_b128 v[{tbase + 2 * k}:{tbase + 2 * k + 3}], %[addr] offset:{8 * row}")
        else:
            L.append(f"ds_read_b64 {t(k)}, %[addr] offset:{8 * row}")
        queue.append(k)

    def consume(k):
        idx = queue.index(k)
        L.append(f"s_waitcnt lgkmcnt({min(len(queue) - 1 - idx, 15)})")
        del queue[: idx + 1]

    for k in range(0, depth, per):
        read(k, k)
    group = max(wait_every, per)
    for i0 in range(0, rows, group):
        # the last read this group needs
        last = (i0 + group - 1) % depth
        consume(last - (last % per))
        for i in range(i0, i0 + group):
            k = i % depth
            L.append(f"v_min_f64 {t(k)}, {t(k)}, %[vn]")
            L.append(f"v_add_f64 %[a{i}], %[a{i}], {t(k)}")
            if i % per == per - 1 and i + depth - (per - 1) < rows:
                read(k - (per - 1), i + depth - (per - 1))
    outs = ("," + BS).join(f'          [a{i}] "+v"(ACC[{i}])' for i in range(rows))
    tmps = ("," + BS).join(f'          "=&{{v[{tbase + 2 * k}:{tbase + 2 * k + 1}]}}"(TMP[{k}])' for k in range(depth))
    return f"""#define {name}(ACC, TMP, ADDR, VN) \\
    asm volatile( \\
{q(L)} \\
        : \\
{outs}, \\
{tmps} \\
        : [addr] "v"(ADDR), [vn] "v"(VN) \\
        : "memory")
"""


L_VARIANTS = [
    dict(b128=False, wait_every=1),
    dict(b128=False, wait_every=4),
    dict(b128=True, wait_every=2),
    dict(b128=True, wait_every=4),
    dict(b128=True, wait_every=8, depth=16),
    dict(b128=False, wait_every=8, depth=16),
    dict(b128=False, wait_every=1, rows=32),   # V6: half a row block per wave (32 accumulators: more waves per SIMD)
    dict(b128=True, wait_every=2, rows=32),    # V7
]

H_VARIANTS = [  # (profiles/r05_cells_probe_rank_halved.txt was collected over several such lists)
    dict(mode="once", group=8, depth=8),
    dict(mode="onoff", group=8, depth=8),
    dict(mode="once", group=4, depth=8),
    dict(mode="once", group=8, depth=8, same_idx=True),
    dict(mode="once", group=8, depth=8, noidx=2),
    dict(mode="once", group=8, depth=8, noidx=2, weave=False),
    dict(mode="once", group=8, depth=8, read2=True),
]


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "tools/cells_probe_asm.h"
    open(path, "w").write("// GENERATED by tools/gen_probe_asm.py -- do not edit.\n#pragma once\n" + probe_d()
                          + f"#define PROBE_H_ACC0 {ACC0}\n#define PROBE_H_VARIANTS {len(H_VARIANTS)}\n"
                          + "".join(probe_h(name=f"PROBE_H_ASM_V{i}", **v) for i, v in enumerate(H_VARIANTS))
                          + "// " + " | ".join(f"V{i}: {v}" for i, v in enumerate(H_VARIANTS)) + "\n"
                          + "".join(probe_l(name=f"PROBE_L_ASM_V{i}", **v) for i, v in enumerate(L_VARIANTS))
                          + "// L: " + " | ".join(f"V{i}: {v}" for i, v in enumerate(L_VARIANTS)) + "\n")
