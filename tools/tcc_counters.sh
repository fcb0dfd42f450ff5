#!/bin/bash
# Runs on the GPU box (under gpurun): L2 hit / miss counters of the tile kernels at configs[2] and configs[3]
# (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum, its own pass: no other trace domains), and a calibration of
# FETCH_SIZE on the symmetric SYMM stream (a known byte count: the tiles it reads once) in the same process.
# usage: tools/tcc_counters.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tcc
rm -rf $out && mkdir -p $out
c2="--steps 2 --warmup 0 --no-cpu-baseline --no-extra --no-parity"
c3="--workload cfg3 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-parity"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc_cfg2 -- python3 bench.py $c2 > /dev/null 2> $out/tcc_cfg2.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc_cfg3 -- python3 bench.py $c3 > /dev/null 2> $out/tcc_cfg3.err
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum --output-format csv -d $out/req_cfg3 -- python3 bench.py $c3 > /dev/null 2> $out/req_cfg3.err || true
for d in tcc_cfg2 tcc_cfg3 req_cfg3; do
  echo "=== $d"
  python3 tools/summarize_counters.py $out/$d | grep -A3 -E "k_accumulate|k_symm_tri|k_block_records|k_sparse" | head -60
done
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -size +30M -delete
