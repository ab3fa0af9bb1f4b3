#!/bin/bash
# Runs ON THE GPU BOX: everything profiles/<tag>_* is made of (tools/summarise_profiles.py <tag> turns it into the committed files)
#   gpurun --timeout 2400 -- 'bash tools/collect_round.sh r02'
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
cd $R
timeout 400 bash tools/prof_engine.sh bench 8 8 | head -12
timeout 400 bash tools/prof_engine.sh solo8 1 8 | head -30
timeout 700 bash tools/pmc_engine.sh $TAG 1 8 > gpurun_out/${TAG}_pmc.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python3 $R/tools/time_train.py --steps 6 > $R/gpurun_out/prof_train.log 2>&1)
timeout 900 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
tail -c 400 gpurun_out/bench_line.json
