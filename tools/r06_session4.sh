cd $GRAFT_REPO_ROOT
export SG_SCENE_CACHE=/tmp/sg_scenes
echo "== big-segment tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_scene.py tests/test_gpu_loader.py -q -x -k "beyond_the_lds or scanned_seed or fresh_full or unusual or walk_modes or loader or group_shape" 2>&1 | tail -5 | cut -c1-300
echo "== ScanNet-shaped solo kernel stats"
timeout 600 bash tools/r05_prof_solo.sh scannet "bigseg|fps|sort_boxes" 2>&1 | tail -12
echo "== bench, ScanNet-shaped profile"
timeout 600 python3 bench.py --seg-profile scannet --steps 20 --repeats 2 --warmup 4 --no-cpu-baseline --no-files --no-extras --parity-scenes 8 --no-oos --scene-cache $SG_SCENE_CACHE 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['roofline']['stage_ms_in_timed_region']
print('scannet', d['repeat_values']['scenes_per_s'], 'parity', d['parity_check']['ranks_equal'], 'fps64 in-region', st.get('fps64'), 'solo', d['roofline']['stage_ms_solo_batched'].get('fps64'))"
timeout 600 python3 bench.py --steps 30 --repeats 2 --warmup 6 --no-cpu-baseline --no-files --no-extras --parity-scenes 8 --no-oos --scene-cache $SG_SCENE_CACHE 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('voronoi', d['repeat_values']['scenes_per_s'], 'parity', d['parity_check']['ranks_equal'])"
echo "== driver end to end, 2048 scenes on tmpfs"
timeout 1200 python3 tools/time_driver.py --scenes 2048 --base /dev/shm --skip-nopack --skip-loop --distinct 32 --out-format "npy@6;txt,npy@8" --out gpurun_out/r06_driver_quick.json > gpurun_out/r06_driver_quick.log 2>&1
python3 -c "
import json; d=json.load(open('gpurun_out/r06_driver_quick.json'))
for k in ('npy@6','txt,npy@8'):
    print(k, {a: (b.get('scenes_per_s'), b.get('steady_scenes_per_s'), b.get('startup_s')) for a,b in d[k].items() if isinstance(b, dict)})"
