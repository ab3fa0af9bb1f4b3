#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 bench.py --generate-only --no-extras --scene-cache /tmp/sg_scenes 2>/dev/null
python3 bench.py --generate-only --no-extras --seg-profile scannet --scene-cache /tmp/sg_scenes 2>/dev/null
for rep in 1 2; do
for G in 10 12 11; do
  python3 bench.py --steps 80 --warmup 10 --repeats 3 --no-cpu-baseline --no-files --no-extras --groups $G --scene-cache /tmp/sg_scenes > gpurun_out/sw.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/sw.json'));print('voronoi groups $G:',d['repeat_values']['scenes_per_s'])"
done
done
for G in 10 12; do
  python3 bench.py --steps 40 --warmup 10 --repeats 3 --no-cpu-baseline --no-files --no-extras --groups $G --seg-profile scannet --scene-cache /tmp/sg_scenes > gpurun_out/sw.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/sw.json'));print('scannet groups $G:',d['repeat_values']['scenes_per_s'])"
done
