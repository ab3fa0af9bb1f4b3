#!/bin/bash
# round-5 helper, runs ON THE GPU BOX:  bash tools/r05_run.sh "<pytest args>" [time_engine profiles...]
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R; mkdir -p gpurun_out
if [ -n "$1" ]; then timeout 1500 python3 -m pytest $1 -x -q 2>&1 | tail -15; fi
shift
for prof in "$@"; do
  python3 tools/time_engine.py --scenes 16 --profile $prof --tag $prof 2>/dev/null | tail -1
done
