#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
make -C seggroup_amd/csrc clean > /dev/null; make -C seggroup_amd/csrc -j 32 PROFILE=1 > gpurun_out/p3_build.log 2>&1; tail -1 gpurun_out/p3_build.log
python3 tools/knn_counters.py voronoi 16 2>&1 | tail -3
python3 tools/knn_counters.py scannet 16 2>&1 | tail -3
