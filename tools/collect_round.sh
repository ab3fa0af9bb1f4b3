#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/<tag>_* is made of, in ONE invocation, ending with the default bench line measured on the
# same code (bench.py reads the profiles/<tag>_pmc_kernels.json this invocation produced for roofline.traffic / knn_valu):
#   gpurun --timeout 3300 -- 'bash tools/collect_round.sh r06'
# What it wrote travels back under gpurun_out/ (profiles_<tag>/ = the files to commit under profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
# scenes of the profiled runs, generated once by an unprofiled process pool: a profiled process must not spawn (the profiler's preload
# has initialised the GPU before python starts)
timeout 600 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE
timeout 500 bash tools/prof_engine.sh bench 14 8 | head -12
timeout 500 bash tools/prof_engine.sh solo8 1 8 | head -30
timeout 900 bash tools/pmc_engine.sh $TAG 1 8 > gpurun_out/${TAG}_pmc.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/tools/time_train.py --steps 6 > $R/gpurun_out/prof_train.log 2>&1)
timeout 600 bash tools/stress_500k.sh $TAG > gpurun_out/${TAG}_stress.log 2>&1
# round 5: the ScanNet-shaped profile's kernels (one group of 8 alone), the driver end to end on tmpfs, the host-side rehearsal of 1 / 2 / 4 / 8 ranks
timeout 600 bash tools/r05_prof_solo.sh scannet . > gpurun_out/${TAG}_solo_scannet.log 2>&1
cp gpurun_out/solo_scannet_kernel_stats.csv profiles/${TAG}_solo_batched_scannet_kernel_stats_raw.csv 2>/dev/null
timeout 900 python3 tools/time_driver.py --scenes 2048 --base /dev/shm --skip-nopack --skip-loop --distinct 32 --out-format "npy@6;txt,npy@8" --out profiles/${TAG}_driver_end_to_end.json > gpurun_out/${TAG}_driver.log 2>&1
timeout 900 python3 tools/host_scale_rehearsal.py --ranks 1,2,4,8 --scenes 768 --rate 3000 --out profiles/${TAG}_host_scale.json > gpurun_out/${TAG}_rehearsal.log 2>&1
# round 6: the bench under roctx ranges (bench.py --profile: kernel trace + marker trace, no counters) and the overlap experiments' table
timeout 400 bash tools/prof_ranges.sh $TAG 14 8 > gpurun_out/${TAG}_ranges.log 2>&1
for k in marker_api_stats kernel_stats; do cp gpurun_out/${TAG}_ranges_${k}.csv profiles/${TAG}_ranges_${k}.csv 2>/dev/null; done
[ -f gpurun_out/r06_overlap.txt ] && cp gpurun_out/r06_overlap.txt profiles/${TAG}_overlap_experiments.txt
[ -f gpurun_out/r06_ec_lever.txt ] && cp gpurun_out/r06_ec_lever.txt profiles/${TAG}_edgeconv_bn2_lever.txt
# exits non-zero (and says why) when a summary averages launches of different sizes: the run goes on, the files say checks.ok = false
python3 tools/summarise_profiles.py $TAG | tail -30
timeout 1200 python3 bench.py --scene-cache $SG_SCENE_CACHE > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
cp gpurun_out/bench_line.json profiles/${TAG}_bench.json
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
tail -c 600 gpurun_out/bench_line.json
