cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_scene.py -x -q -k "concurrent or group_shape or tiny" 2>&1 | tail -3
bash tools/r06_overlap.sh 2>&1 | tail -40
export SG_SCENE_CACHE=/tmp/sg_scenes
timeout 400 bash tools/prof_engine.sh solo8 1 8 | head -14
timeout 300 bash tools/prof_ranges.sh r06 10 8 | tail -50
