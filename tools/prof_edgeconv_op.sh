#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of tools/time_edgeconv.py (the EdgeConv operator alone, both slot loops).
#   gpurun --timeout 600 -- 'bash tools/prof_edgeconv_op.sh [tag] [N]'
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ecop}; N=${2:-150000}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/time_edgeconv.py $N 20 > $R/gpurun_out/prof_$TAG.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n=r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    print("%-50s calls %5s total %9.3f ms avg %9.2f us"%(n[:50],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3))
PY
grep -v "^W2026\|^E2026" $R/gpurun_out/prof_$TAG.log | tail -12 | cut -c1-200
