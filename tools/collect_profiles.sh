#!/bin/bash
# Runs on the GPU box (under gpurun): kernel-time profile, the two PMC passes and the plain
# default bench line of BASELINE.json configs[2], all into gpurun_out/profiles_raw/.
# usage: tools/collect_profiles.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_raw
rm -rf $out && mkdir -p $out
args="--steps 5 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py $args > $out/bench_under_rocprof.json 2> $out/kt.err
echo "kernel trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $out/fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $out/write.err
echo "WRITE_SIZE pass done"
SCS_ACC_STAMP=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $out/stamps.txt || true
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "default bench done"
# keep only the small files
find $out -name "*kernel_trace.csv" -size +20M -delete
ls -la $out $out/*/* | head -40
