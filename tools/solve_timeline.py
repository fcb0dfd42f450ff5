#!/usr/bin/env python3
"""Timeline of the LAST Fiedler solve in a rocprofv3 --kernel-trace csv (of `bench.py --steps K`):
every launch after the last accumulate kernel, with its start offset, duration and the idle gap in
front of it -- the head and the tail of the solve in full, the iterations in between as one average.

usage: tools/solve_timeline.py <dir with *_kernel_trace.csv> [head=14] [tail=26]"""
import csv
import glob
import sys


def main() -> None:
    root = sys.argv[1]
    head = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    tail = int(sys.argv[3]) if len(sys.argv) > 3 else 26
    path = sorted(glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True))[-1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    last_acc = max(i for i, r in enumerate(rows) if "k_accumulate" in r[2] or "k_tp_" in r[2])
    solve = rows[last_acc + 1:]
    t0 = rows[last_acc][1]
    print(f"{len(solve)} launches after the last accumulate kernel; total {1e-3 * (solve[-1][1] - t0):.1f} us")
    prev = t0
    out = []
    for s, e, name in solve:
        out.append((1e-3 * (s - t0), 1e-3 * (e - s), 1e-3 * (s - prev), name[:90]))
        prev = e
    def show(items):
        for off, dur, gap, name in items:
            print(f"  +{off:9.1f} us  dur {dur:7.1f}  gap {gap:6.1f}  {name}")
    show(out[:head])
    mid = out[head:len(out) - tail]
    if mid:
        busy = sum(d for _, d, _, _ in mid)
        gaps = sum(g for _, _, g, _ in mid)
        print(f"  ... {len(mid)} launches: busy {busy:.1f} us, gaps {gaps:.1f} us")
        by = {}
        for _, d, g, name in mid:
            k = by.setdefault(name, [0, 0.0, 0.0])
            k[0] += 1
            k[1] += d
            k[2] += g
        for name, (c, d, g) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            print(f"      {c:4d} x {d / c:7.1f} us (+ gap {g / c:5.1f})  {name}")
    show(out[len(out) - tail:])


if __name__ == "__main__":
    main()
