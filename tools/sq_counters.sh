#!/bin/bash
# Runs on the GPU box (under gpurun): SQ counter passes over the configs[2] bench for the
# accumulate kernel (VALU / LDS busy, bank conflicts, waits), into gpurun_out/$1/.
# usage: tools/sq_counters.sh <tag> [bench args]
set -e
tag=${1:-sq}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
args="--steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-parity $@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/p1 -- python3 bench.py $args > $out/p1.json 2> $out/p1.err
echo "pass 1 done"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/p2 -- python3 bench.py $args > $out/p2.json 2> $out/p2.err
echo "pass 2 done"
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS GRBM_GUI_ACTIVE --output-format csv -d $out/p3 -- python3 bench.py $args > $out/p3.json 2> $out/p3.err || echo "pass 3 failed (counter names?)"
echo "pass 3 done"
python3 tools/summarize_counters.py $out > $out/summary.txt || true
cat $out/summary.txt
find $out -name "*kernel_trace.csv" -size +20M -delete
