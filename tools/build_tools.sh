#!/bin/bash
# builds the tuning harnesses next to their sources (not part of the library)
set -e
here="$(cd "$(dirname "$0")" && pwd)"
for t in symm_bench fma_probe; do
  [ -f "$here/$t.hip" ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$here/$t.hip" -o "$here/$t"
done
echo built
