cd $GRAFT_REPO_ROOT
export SG_SCENE_CACHE=/tmp/sg_scenes
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
run() { name=$1; shift; env "$@" timeout 300 python3 bench.py --steps 60 --repeats 3 --warmup 8 --no-cpu-baseline --no-files --no-extras --parity-scenes 16 --scene-cache $SG_SCENE_CACHE $EXTRA 2>gpurun_out/s16_$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-22s' % '$name', d['repeat_values']['scenes_per_s'], d['parity_check']['ranks_equal'], d['parity_check']['out_of_step']['wrong_on_rank0'], d['engine_profile'])
except Exception as e: print('$name FAILED', e)"; }
for rep in 1 2; do
EXTRA="" run sdma SG_X=0
EXTRA="" run hip SG_ENGINE_LABEL_COPY=hip
EXTRA="--label-transfer tables" run tables SG_X=0
done
tail -3 gpurun_out/s16_sdma.err
