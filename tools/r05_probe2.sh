#!/bin/bash
# is the ScanNet-shaped profile short of GPU work or of latency cover?  same 64 scenes, 10 / 14 / 18 engine groups
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
mkdir -p gpurun_out
timeout 600 python3 bench.py --generate-only --no-extras --seg-profile scannet --scene-cache $SG_SCENE_CACHE
timeout 600 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE
for G in 10 14 18; do
  timeout 600 python3 bench.py --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-files --no-extras --groups $G --seg-profile scannet --scene-cache $SG_SCENE_CACHE > gpurun_out/p2_scannet_g$G.json 2> gpurun_out/p2_scannet_g$G.err
  python3 -c "import json;d=json.load(open('gpurun_out/p2_scannet_g$G.json'));print('scannet groups',$G,d['value'],d['repeat_values']['scenes_per_s'],d['engine_profile'])"
done
for G in 10 14; do
  timeout 600 python3 bench.py --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-files --no-extras --groups $G --scene-cache $SG_SCENE_CACHE > gpurun_out/p2_voronoi_g$G.json 2> gpurun_out/p2_voronoi_g$G.err
  python3 -c "import json;d=json.load(open('gpurun_out/p2_voronoi_g$G.json'));print('voronoi groups',$G,d['value'],d['repeat_values']['scenes_per_s'],d['engine_profile'])"
done
