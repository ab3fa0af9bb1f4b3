R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export SG_SCENE_CACHE=/tmp/sg_scenes
timeout 600 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE > /dev/null 2>&1
T() { timeout 300 python3 tools/time_engine.py --scene-cache $SG_SCENE_CACHE --tag $1 2>&1 | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); u=d['us_per_scene']; print(d['tag'], d['scenes_per_s'], 'S1X', u['kernel.l2.edgeconv'], 'S2X', u['kernel.l3.edgeconv'], 'sum', d['sum_us_per_scene'])"; }
timeout 900 python3 -m pytest tests -x -q -m gpu -k "edgeconv or labels_and_metrics or 150k_scene or batch_of_64 or scanned_seed" 2>&1 | tail -1
T warm >/dev/null
for rep in 1 2 3; do SEGGROUP_HIP_LIB=$R/build_micro/lib_walk_strided.so T before; T after; done
