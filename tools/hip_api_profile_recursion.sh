cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/hipapi_rec && mkdir -p gpurun_out/hipapi_rec
rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/hipapi_rec -- python3 tools/node_phases.py 8000 1000 > gpurun_out/hipapi_rec/phases.json 2> gpurun_out/hipapi_rec/err.txt
f=$(ls gpurun_out/hipapi_rec/*/*hip_api_stats.csv | head -1); head -16 "$f"
find gpurun_out/hipapi_rec -name "*trace.csv" -delete
python3 -c "
import json;d=json.load(open('gpurun_out/hipapi_rec/phases.json'));print(d['total_s'])"
