cd $GRAFT_REPO_ROOT
echo "== loader / cache / ranks tests"
timeout 1500 python -m pytest tests/test_gpu_loader.py tests/test_gpu_ranks.py -q -x 2>&1 | tail -4 | cut -c1-300
echo "== driver sweep: uploads on the copy engines (default) vs hipMemcpyAsync"
timeout 1500 python3 tools/sweep_driver.py sdma hip,SG_LOADER_COPY=hip sdma_thr6,SG_LOADER_THREADS=6 sdma_thr8,SG_LOADER_THREADS=8 sdma_copies4,SG_LOADER_COPIES=4 2>&1 | grep "overall" | tail -12
echo "== txt,npy"
timeout 1500 python3 tools/sweep_driver.py --format txt,npy --workers 8 sdma hip,SG_LOADER_COPY=hip 2>&1 | grep "overall" | tail -6
