cd $GRAFT_REPO_ROOT
timeout 1500 python3 tools/parity_sweep.py --tag r06 --full 16 --workers 32 2>&1 | tail -3
cp profiles/r06_parity_sweep.json gpurun_out/ 2>/dev/null
export SG_SCENE_CACHE=/tmp/sg_scenes
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
for t in 1 0 1 0; do
timeout 300 python3 bench.py --engine-timing $t --steps 40 --repeats 2 --warmup 6 --no-cpu-baseline --no-files --no-extras --parity-scenes 8 --no-oos --scene-cache $SG_SCENE_CACHE 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('engine-timing $t', d['repeat_values']['scenes_per_s'], d['parity_check']['ranks_equal'], len(d['roofline']['stage_ms_in_timed_region']))"
done
