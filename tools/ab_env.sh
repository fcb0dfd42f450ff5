#!/bin/bash
# A/B of one environment switch on one box: tools/ab_env.sh VAR=a VAR=b -- bench args
a="$1"; b="$2"; shift 3
for round in 1 2; do
  for v in "$a" "$b"; do
    env "$v" python bench.py --no-extra --no-cpu-baseline "$@" > gpurun_out/abenv.json 2> gpurun_out/abenv.err
    python - "$v" <<'PY'
import json,sys
r=json.load(open("gpurun_out/abenv.json")); s=r["stages"]
print(sys.argv[1], r["value"], "build", s["build_ms"], "acc", s["build_accumulate_ms"], "solve", s["fiedler_ms"], "mismatch", r["parity"]["w_cells_mismatched"])
PY
  done
done
