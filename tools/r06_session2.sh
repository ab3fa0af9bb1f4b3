cd $GRAFT_REPO_ROOT
echo "== gate check: the list-form kNN WITHOUT the score_bound fix must fail the new gates"
SEGGROUP_HIP_LIB=$PWD/build_micro/libsg_listform_unfixed.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "slices_short" 2>&1 | tail -8
SEGGROUP_HIP_LIB=$PWD/build_micro/libsg_listform_unfixed.so timeout 600 python -m pytest tests/test_gpu_scene.py -q -x -k "small_scenes_engine" 2>&1 | tail -5
echo "== full GPU suite on the tree"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
export SG_SCENE_CACHE=/tmp/sg_scenes
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
bash tools/r06_overlap.sh "base,SG_X=0" "walk1.75,SG_EC_WALK=1.75" "base,SG_X=0" "walk1.75,SG_EC_WALK=1.75" 2>&1 | tail -4
