#!/bin/bash
# Runs ON THE GPU BOX: scenes/s of the bench for a few engine shapes (groups x scenes per batched launch); scenes come from a cache filled once
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
if [ $# -eq 0 ]; then set -- "8 8" "10 8" "12 8" "8 12" "6 12" "16 4" "12 6"; fi
for shape in "$@"; do
  set -- $shape
  python3 bench.py --groups $1 --per-group $2 --steps ${SG_SWEEP_STEPS:-40} --repeats ${SG_SWEEP_REPEATS:-2} --no-extras --no-files --no-cpu-baseline --parity-scenes 8 --scene-cache $SG_SCENE_CACHE 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 x $2', d['repeat_values']['scenes_per_s'], d['engine_profile'])"
done
