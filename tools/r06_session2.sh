cd $GRAFT_REPO_ROOT
echo "== gate check: the list-form kNN WITH the round-5 NaN bound must fail the gates"
SEGGROUP_HIP_LIB=$PWD/build_micro/libsg_knn_r5_nan_bound.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "slices_short" 2>&1 | grep -E "^FAILED|passed|failed|rows differ|outside the scene" | cut -c1-220 | head -20
SEGGROUP_HIP_LIB=$PWD/build_micro/libsg_knn_r5_nan_bound.so timeout 600 python -m pytest tests/test_gpu_scene.py -q -k "small_scenes_engine" 2>&1 | grep -E "^FAILED|passed|failed|differ" | cut -c1-220 | head
echo "== full GPU suite on the tree"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12
