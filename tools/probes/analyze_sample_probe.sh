#!/bin/bash
# Probe (round 6): the union-find kernel of large universes with a first pass over one wave of leaves in S
# (SCS_ANALYZE_SAMPLE=S; 1 = one pass): durations of the two launches from a kernel trace of a 20 000 x 5 000 recursion.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in 1 64 256 1024 4096; do
rm -rf gpurun_out/twice && mkdir -p gpurun_out/twice
SCS_ANALYZE_SAMPLE=$S rocprofv3 --kernel-trace --output-format csv -d gpurun_out/twice/kt -- python3 tools/levels_profile.py 20000 5000 branch --no-profile > gpurun_out/twice/run.log 2>&1
f=$(find gpurun_out/twice/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" $S <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_analyze_leaves<false>" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
first=d[0::2] if sys.argv[2] != "1" else d; second=d[1::2] if sys.argv[2] != "1" else d
print("sample", sys.argv[2], len(d), "launches; first pass: avg %.0f us (min %.0f max %.0f); full pass: avg %.0f us (min %.0f max %.0f)" % (sum(first)/len(first), min(first), max(first), sum(second)/len(second), min(second), max(second)))
PY
find gpurun_out/twice -name "*kernel_trace.csv" -delete
done
