#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: usage summarize_counters.py <dir>."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(root.rglob("*counter_collection.csv")):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0][:60]
            c = row.get("Counter_Name")
            v = float(row.get("Counter_Value", 0) or 0)
            tot[k][c] += v
            cnt[k][c] += 1
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("SQ_WAIT_ANY", 0))):
    print(f"== {k}")
    for c in sorted(tot[k]):
        print(f"   {c:28s} sum {tot[k][c]:.6g}  dispatches {cnt[k][c]}  per-dispatch {tot[k][c] / max(cnt[k][c], 1):.6g}")
