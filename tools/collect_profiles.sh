#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats of the bench command + PMC HBM-traffic passes.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# Outputs land under gpurun_out/<tag>_*; tools/summarise_profiles.py (run in the build container) turns them
# into the committed files under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the SAME command as the bench line (no CPU leg, no file leg: they launch no kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_bench_stats -- python3 $R/bench.py --no-cpu-baseline --no-files --steps 4 --warmup 1 > $R/gpurun_out/${TAG}_bench_stats.log 2>&1
# 2. single-stream kernel durations (no inter-stream interference)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_solo_stats -- python3 $R/tools/time_scene.py 150000 1500 > $R/gpurun_out/${TAG}_solo_stats.log 2>&1
# 3. HBM traffic: separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- python3 $R/tools/time_scene.py 150000 1500 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- python3 $R/tools/time_scene.py 150000 1500 > /dev/null 2>&1
# 4. stress scene (BASELINE.json configs[4]): 500k points / 5k segments, one scene at a time
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stress_stats -- python3 $R/tools/time_scene.py 500000 5000 > $R/gpurun_out/${TAG}_stress_stats.log 2>&1
# 5. SQ counters per kernel (instruction mix, issue / wait split): one PMC pass, kernel-trace only
bash $R/tools/pmc_kernel.sh ${TAG}_pmc_sq > $R/gpurun_out/${TAG}_pmc_sq.txt 2>&1
tail -1 $R/gpurun_out/${TAG}_bench_stats.log | cut -c1-200
