#!/bin/bash
# Runs on the GPU box (under gpurun): per-kernel times of one whole recursion with the level-synchronous
# engine (rocprofv3 --kernel-trace --stats), top kernels printed.
# usage: tools/levels_kstats.sh TAXA TREES [STRATEGY]   (SCS_SPEC_MAX_TAXA etc. from the environment)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/lkstats
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/levels_profile.py "$@" --no-profile > $out/run.log 2> $out/kt.err
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
grep "taxa /" $out/run.log
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e9:.3f} s")
for r in rows[:32]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:10.1f} us {float(r['TotalDurationNs'])/1e6:9.1f} ms {float(r['Percentage']):6.2f} %")
P
cp "$f" $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
