cd $GRAFT_REPO_ROOT
export SG_SCENE_CACHE=/tmp/sg_scenes
echo "== scene-level gate with the round-5 NaN bound (expected to fail)"
SEGGROUP_HIP_LIB=$PWD/build_micro/libsg_knn_r5_nan_bound.so timeout 900 python -m pytest tests/test_gpu_scene.py -q -x -k "small_scenes_engine" 2>&1 | tail -4 | cut -c1-250
echo "== out-of-step + selfcheck tests"
timeout 1800 python -m pytest tests/test_gpu_scene.py -q -k "group_shape or selfcheck" 2>&1 | tail -6 | cut -c1-300
echo "== EdgeConv lever"
bash tools/r06_ec_lever.sh 2>&1 | tail -9
echo "== solo batched stage times (release library)"
python3 tools/time_engine.py --rounds 3 2>/dev/null | tail -1
echo "== driver end to end, 1024 scenes on tmpfs"
timeout 900 python3 tools/time_driver.py --scenes 1024 --base /dev/shm --skip-nopack --skip-loop --distinct 32 --out-format "npy@6;txt,npy@8" --out gpurun_out/r06_driver_quick.json > gpurun_out/r06_driver_quick.log 2>&1
python3 -c "
import json; d=json.load(open('gpurun_out/r06_driver_quick.json'))
for k in ('npy@6','txt,npy@8'):
    print(k, {a: (b.get('scenes_per_s'), b.get('steady_scenes_per_s'), b.get('startup_s')) for a,b in d[k].items() if isinstance(b, dict)})"
