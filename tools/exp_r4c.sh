#!/bin/bash
# Runs ON THE GPU BOX (round 4): MLP1 apply with packed FMAs (new library vs build_micro/lib_ec_base.so), the layout kernel's block size.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export SG_SCENE_CACHE=/tmp/sg_scenes
timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
T() { timeout 300 python3 tools/time_engine.py --scene-cache $SG_SCENE_CACHE --tag $1 2>&1 | grep '^{' | cut -c1-600; }
T warm > /dev/null
for rep in 1 2; do
  echo "== old lib"; SEGGROUP_HIP_LIB=$R/build_micro/lib_ec_base.so T old
  echo "== new lib"; T new
  for b in 128 64; do echo "== layout block $b"; SG_LAYOUT_BLOCK=$b T layout$b; done
done
