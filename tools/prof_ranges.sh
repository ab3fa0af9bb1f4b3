#!/bin/bash
# Runs ON THE GPU BOX: the bench under rocprofv3 with roctx ranges (bench.py --profile: the legs of the bench from Python, every phase and stage
# of the engine from its group threads), kernel trace + marker trace, no counters.
#   gpurun -- 'bash tools/prof_ranges.sh r06 10 8'   -> gpurun_out/<tag>_ranges_{marker,kernel}_stats.csv (+ the domain summary)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}; G=${2:-10}; B=${3:-8}; CACHE=${SG_SCENE_CACHE:-/tmp/sg_scenes}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $R/gpurun_out/prof_ranges -- python3 $R/bench.py --profile --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-files --groups $G --per-group $B --parity-scenes 0 --no-extras --gen-workers 1 --scene-cache $CACHE > $R/gpurun_out/prof_ranges.log 2>&1
for kind in marker_api_stats kernel_stats domain_stats; do
  f=$(find $R/gpurun_out/prof_ranges -name "*${kind}.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/${TAG}_ranges_${kind}.csv
done
head -40 $R/gpurun_out/${TAG}_ranges_marker_api_stats.csv 2>/dev/null | cut -c1-160
ls $R/gpurun_out/prof_ranges/*/ 2>/dev/null | head
tail -c 300 $R/gpurun_out/prof_ranges.log
rm -rf $R/gpurun_out/prof_ranges
