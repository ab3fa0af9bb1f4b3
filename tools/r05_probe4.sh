#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
mkdir -p gpurun_out
timeout 600 python3 bench.py --generate-only --no-extras --seg-profile scannet --scene-cache $SG_SCENE_CACHE
timeout 600 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE
for P in scannet voronoi; do
  timeout 600 python3 bench.py --steps 40 --warmup 5 --repeats 2 --no-cpu-baseline --no-files --no-extras --seg-profile $P --scene-cache $SG_SCENE_CACHE > gpurun_out/p4_$P.json 2> gpurun_out/p4_$P.err
  python3 -c "import json;d=json.load(open('gpurun_out/p4_$P.json'));print('$P',d['value'],d['repeat_values']['scenes_per_s'],d['engine_profile'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scannet -- python3 $R/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-files --groups 1 --per-group 8 --no-extras --gen-workers 1 --seg-profile scannet --batch 16 --scene-cache $SG_SCENE_CACHE > $R/gpurun_out/prof_scannet.log 2>&1
f=$(find $R/gpurun_out/prof_scannet -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/p4_scannet_kernel_stats.csv
rm -rf $R/gpurun_out/prof_scannet
head -24 $R/gpurun_out/p4_scannet_kernel_stats.csv | cut -c1-110
