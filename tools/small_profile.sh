cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/smallprof && mkdir -p gpurun_out/smallprof
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/smallprof -- python3 tools/node_phases.py 6000 1000 > gpurun_out/smallprof/phases.json 2> gpurun_out/smallprof/err.txt
f=$(ls gpurun_out/smallprof/*/*kernel_stats.csv | head -1); head -12 $f
g=$(ls gpurun_out/smallprof/*/*memory_copy_stats.csv | head -1); cat $g
find gpurun_out/smallprof -name "*trace.csv" -size +5M -delete
python3 - <<'P'
import json;d=json.load(open('gpurun_out/smallprof/phases.json'));print(d['total_s'],{k:v for k,v in d['phases'].items() if k in('small_solve','split','flatten','kmeans')})
P
