#!/bin/bash
# Runs on the GPU box (under gpurun): per-kernel times of the default bench step only
# (rocprofv3 --kernel-trace --stats), top kernels printed -- the quick look between two
# kernel changes; tools/collect_profiles.sh is the full evidence run.
# usage: tools/kernel_stats.sh [extra bench.py arguments]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/kstats
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --steps 5 --no-cpu-baseline --no-extra --no-parity "$@" > $out/bench.json 2> $out/kt.err
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:64]:64s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.1f} us {float(r['Percentage']):6.2f} %")
P
cp "$f" $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete
python3 -c "
import json; d = json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
