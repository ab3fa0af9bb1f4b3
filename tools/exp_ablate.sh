# Development only: build the patched kernel first (python tools/exp_ablate_patch.py && make -C seggroup_amd/csrc), undo with git checkout afterwards
# (the patch applies to the slot loop of commit 55849f3: see its header)
export SG_SCENE_CACHE=/tmp/sg_scenes
python3 bench.py --generate-only --no-extras --batch 16 --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
for fl in 0 1 2 4 8 16 32 3 7 63; do SG_EC_STAGGER2=$((fl*256)) python3 tools/time_engine.py --tag ablate=$fl --rounds 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['tag'], 'S2X', d['us_per_scene']['kernel.l3.edgeconv'], 'S1X', d['us_per_scene']['kernel.l2.edgeconv'])"; done
