#!/bin/bash
# Runs ON THE GPU BOX: BASELINE.json configs[4] -- one 500k-point / 5k-segment scene at a time through the single-scene pipeline:
# rocprofv3 kernel stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains), then
# tools/stress_report.py writes profiles/<tag>_stress_500k_report.md (per kernel: us, HBM bytes, GB/s or MFMA TFLOP/s against its roof).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}; N=${2:-500000}; S=${3:-5000}
O=$R/gpurun_out/${TAG}_stress
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_stats -- python3 $R/tools/time_scene.py $N $S 6 > ${O}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${O}_fetch -- python3 $R/tools/time_scene.py $N $S 4 > ${O}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${O}_write -- python3 $R/tools/time_scene.py $N $S 4 > ${O}_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d ${O}_valu -- python3 $R/tools/time_scene.py $N $S 4 > ${O}_valu.log 2>&1
cd $R && python3 tools/stress_report.py $TAG $N $S
