#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for GB in "8 8" "10 8" "12 8" "14 8" "10 6" "13 6" "16 5" "10 10" "8 10"; do
  set -- $GB
  python3 bench.py --steps 40 --warmup 5 --repeats 2 --no-cpu-baseline --no-files --no-extras --groups $1 --per-group $2 --scene-cache /tmp/sg_scenes > gpurun_out/sw.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/sw.json'));print('groups $1 x $2:',d['repeat_values']['scenes_per_s'],{k:round(v,2) for k,v in d['engine_profile'].items()})"
done
