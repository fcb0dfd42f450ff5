#!/bin/bash
# A/B timing of two builds of libscs_hip.so on one box: tools/ab/libB.so against the in-tree one.
# usage: tools/ab.sh [bench args]
set -e
L=spectralclustersupertree_amd/libscs_hip.so
cp $L /tmp/libA.so
for round in 1 2; do
  for v in A B; do
    if [ $v = A ]; then cp /tmp/libA.so $L; else cp tools/ab/libB.so $L; fi
    python bench.py --no-extra --no-cpu-baseline "$@" > gpurun_out/ab_$v$round.json 2> gpurun_out/ab_$v$round.err
    python - <<PY
import json
r=json.load(open("gpurun_out/ab_$v$round.json"))
print("$v$round", r["value"], "build", r["stages"]["build_ms"], "acc", r["stages"]["build_accumulate_ms"], "solve", r["stages"]["fiedler_ms"], "mismatch", r["parity"]["w_cells_mismatched"])
PY
  done
done
cp /tmp/libA.so $L
