"""Last launches of a rocprofv3 --kernel-trace csv with the idle gap in front of each.
    python tools/trace_tail.py <dir> [count]"""
import csv, glob, re, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(path)):
    m = re.search(r"(k_\w+|__amd_\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:60]))
rows.sort()
tail = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 60):]
prev = tail[0][0]
for s, e, n in tail:
    print(f"gap {1e-3*(s-prev):7.1f} dur {1e-3*(e-s):6.1f}  {n}")
    prev = e
