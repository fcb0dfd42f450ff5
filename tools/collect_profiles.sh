#!/bin/bash
# Runs on the GPU box (under gpurun): the round's evidence for profiles/ --
#   kernel-time profile of the default bench step (rocprofv3 --kernel-trace --stats),
#   FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, no other trace domains),
#   SQ counter passes for the accumulate kernel, its in-kernel phase stamps,
#   and the plain default bench line -- all into gpurun_out/profiles_raw/.
# usage: tools/collect_profiles.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/profiles_raw
rm -rf $out && mkdir -p $out
args="--steps 5 --no-cpu-baseline --no-extra --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py $args > $out/bench_under_rocprof.json 2> $out/kt.err
echo "kernel trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra --no-parity > /dev/null 2> $out/fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra --no-parity > /dev/null 2> $out/write.err
echo "WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq1 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra --no-parity > /dev/null 2> $out/sq1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out/sq2 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra --no-parity > /dev/null 2> $out/sq2.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/sq3 -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extra --no-parity > /dev/null 2> $out/sq3.err
echo "SQ passes done"
SCS_DEBUG=1 SCS_ACC_STAMP=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-parity > /dev/null 2> $out/stamps.txt || true
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "default bench done"
# configs[3] on one device and configs[2]'s shape under `bootstrap` (k_accumulate_gen): kernel times and traffic
c3="--workload cfg3 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_cfg3 -- python3 bench.py $c3 > $out/bench_cfg3_under_rocprof.json 2> $out/kt_cfg3.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_cfg3 -- python3 bench.py $c3 > /dev/null 2> $out/fetch_cfg3.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write_cfg3 -- python3 bench.py $c3 > /dev/null 2> $out/write_cfg3.err
echo "cfg3 passes done"
bs="--workload custom --taxa 10000 --trees 500 --strategy bootstrap --steps 2 --warmup 0 --no-cpu-baseline --no-extra --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_boot -- python3 bench.py $bs > $out/bench_boot_under_rocprof.json 2> $out/kt_boot.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_boot -- python3 bench.py $bs > /dev/null 2> $out/fetch_boot.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write_boot -- python3 bench.py $bs > /dev/null 2> $out/write_boot.err
echo "bootstrap passes done"
python3 bench.py --workload cfg4 --steps 1 --no-extra --no-cpu-baseline > $out/bench_cfg4.json 2> $out/bench_cfg4.err || true
echo "cfg4 done"
python3 tools/summarize_counters.py $out > $out/counters_summary.txt || true
# keep only the small files
find $out -name "*kernel_trace.csv" -size +20M -delete
ls -la $out | head -40
