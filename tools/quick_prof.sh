#!/bin/bash
# Runs ON THE GPU BOX: the solo-batched kernel table (one engine group of 8 scenes alone on the GPU) for a quick before / after.
#   gpurun --timeout 900 -- 'bash tools/quick_prof.sh [tag]'
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
timeout 300 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE
timeout 500 bash tools/prof_engine.sh ${1:-solo8} 1 8 | head -${2:-34}
