# HIP API calls of the protocol steps (rocprofv3 --hip-trace --stats; run on the GPU box): which
# runtime calls a step makes and what they cost the host.   usage: tools/hip_api_profile.sh [taxa] [trees]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/hipapi && mkdir -p gpurun_out/hipapi
rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/hipapi -- python3 tools/overhead_check.py ${1:-10000} ${2:-500} > gpurun_out/hipapi/steps.txt 2> gpurun_out/hipapi/err.txt
f=$(ls gpurun_out/hipapi/*/*hip_api_stats.csv | head -1); head -30 "$f"
find gpurun_out/hipapi -name "*trace.csv" -size +5M -delete
tail -3 gpurun_out/hipapi/steps.txt
