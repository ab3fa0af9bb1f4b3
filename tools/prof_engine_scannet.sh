#!/bin/bash
# Runs ON THE GPU BOX: like prof_engine.sh (1 group x 8: batched launches alone on the GPU) on the ScanNet-shaped segment profile
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-solo8_scannet}; G=${2:-1}; B=${3:-8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --batch 16 --no-cpu-baseline --no-files --groups $G --per-group $B --no-extras --seg-profile scannet --parity-scenes 0 --repeats 1 --gen-workers 1 --scene-cache ${SG_SCENE_CACHE:-/tmp/sg_scenes} > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    n=r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    print("%-50s calls %5s total %9.3f ms avg %9.2f us  %5.1f%%"%(n[:50],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
tail -c 400 $R/gpurun_out/prof_$TAG.log | cut -c1-300
