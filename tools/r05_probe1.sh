#!/bin/bash
# round 5, first GPU trip: new tests + where the ScanNet-shaped profile's time goes (rocprof kernel stats, solo batched)
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_ranks.py -x -q -k "full_size or rccl" > gpurun_out/p1_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/p1_tests.log
timeout 600 python3 bench.py --generate-only --no-extras --seg-profile scannet --scene-cache $SG_SCENE_CACHE --batch 16
timeout 600 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE --batch 16
python3 tools/time_engine.py --scenes 16 --profile scannet --tag scannet > gpurun_out/p1_te_scannet.json 2> gpurun_out/p1_te_scannet.err
python3 tools/time_engine.py --scenes 16 --profile voronoi --tag voronoi > gpurun_out/p1_te_voronoi.json 2> gpurun_out/p1_te_voronoi.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scannet -- python3 $R/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-files --groups 1 --per-group 8 --no-extras --gen-workers 1 --seg-profile scannet --batch 16 --scene-cache $SG_SCENE_CACHE > $R/gpurun_out/prof_scannet.log 2>&1
f=$(find $R/gpurun_out/prof_scannet -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/p1_scannet_kernel_stats.csv
rm -rf $R/gpurun_out/prof_scannet
cd $R
cat gpurun_out/p1_te_scannet.json gpurun_out/p1_te_voronoi.json
head -30 gpurun_out/p1_scannet_kernel_stats.csv | cut -c1-160
tail -3 gpurun_out/p1_tests.log
