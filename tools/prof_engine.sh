#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the bench with a given engine shape (default 1 group x 8: batched launches, nothing else on the GPU).
# The profiled process spawns nothing (--gen-workers 1): its scenes come from the cache tools/collect_round.sh filled beforehand.
# --parity-scenes 0: no parity legs inside the profiled process (round 5's out-of-step leg launched the SAME batched kernels with one scene each and
# diluted every per-launch average of the summary; tools/summarise_profiles.py now refuses a summary whose launches are not all full).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-solo8}; G=${2:-1}; B=${3:-8}; CACHE=${SG_SCENE_CACHE:-/tmp/sg_scenes}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-files --groups $G --per-group $B --parity-scenes 0 --no-extras --gen-workers 1 --scene-cache $CACHE > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:32]:
    n=r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    print("%-50s calls %5s total %9.3f ms avg %9.2f us  %5.1f%%"%(n[:50],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot))
PY
tail -c 600 $R/gpurun_out/prof_$TAG.log | cut -c1-300
