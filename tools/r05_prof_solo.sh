#!/bin/bash
# rocprofv3 kernel stats of ONE engine group (8 scenes per batched launch) alone on the GPU:  bash tools/r05_prof_solo.sh <voronoi|scannet> [grep pattern]
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-voronoi}; PAT=${2:-.}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R; mkdir -p gpurun_out
timeout 600 python3 bench.py --generate-only --no-extras --seg-profile $P --scene-cache $SG_SCENE_CACHE --batch 16 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solo_$P -- python3 $R/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-files --groups 1 --per-group 8 --parity-scenes 0 --no-extras --gen-workers 1 --seg-profile $P --batch 16 --scene-cache $SG_SCENE_CACHE > $R/gpurun_out/prof_solo_$P.log 2>&1
f=$(find $R/gpurun_out/prof_solo_$P -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/solo_${P}_kernel_stats.csv
rm -rf $R/gpurun_out/prof_solo_$P
python3 - $R/gpurun_out/solo_${P}_kernel_stats.csv "$PAT" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n=r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    if re.search(sys.argv[2],n) and "_b" in n:
        print("%-46s calls %4s avg %9.2f us min %9.2f max %9.2f"%(n[:46],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
