export SG_SCENE_CACHE=/tmp/sg_scenes
python3 bench.py --generate-only --no-extras --batch 16 --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
python3 tools/time_engine.py --tag ${1:-x} --rounds 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['tag'], {k: d['us_per_scene'][k] for k in ('l2.knn','l3.knn','kernel.l2.edgeconv','kernel.l3.edgeconv','mlp1','fps64')})"
