export SG_SCENE_CACHE=/tmp/sg_scenes
python3 bench.py --generate-only --no-extras --batch 16 --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
for s2 in 0 6 12 18 24 32; do SG_EC_STAGGER2=$s2 SG_EC_STAGGER1=0 python3 tools/time_engine.py --tag s2=$s2 2>/dev/null | tail -1; done
for s1 in 3 6 9 12; do SG_EC_STAGGER1=$s1 python3 tools/time_engine.py --tag s1=$s1 2>/dev/null | tail -1; done
