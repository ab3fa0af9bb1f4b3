#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/<tag>_* is made of, in ONE invocation, ending with the default bench line measured on the
# same code (bench.py reads the profiles/<tag>_pmc_kernels.json this invocation produced for roofline.traffic / knn_valu):
#   gpurun --timeout 2700 -- 'bash tools/collect_round.sh r03'
# What it wrote travels back under gpurun_out/ (profiles_<tag>/ = the files to commit under profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
# scenes of the profiled runs, generated once by an unprofiled process pool: a profiled process must not spawn (the profiler's preload
# has initialised the GPU before python starts)
timeout 600 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE
timeout 500 bash tools/prof_engine.sh bench 10 8 | head -12
timeout 500 bash tools/prof_engine.sh solo8 1 8 | head -30
timeout 900 bash tools/pmc_engine.sh $TAG 1 8 > gpurun_out/${TAG}_pmc.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/tools/time_train.py --steps 6 > $R/gpurun_out/prof_train.log 2>&1)
timeout 600 bash tools/stress_500k.sh $TAG > gpurun_out/${TAG}_stress.log 2>&1
python3 tools/summarise_profiles.py $TAG | tail -30
timeout 1200 python3 bench.py --scene-cache $SG_SCENE_CACHE > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
cp gpurun_out/bench_line.json profiles/${TAG}_bench.json
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
tail -c 600 gpurun_out/bench_line.json
