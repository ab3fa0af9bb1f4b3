#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of tools/time_prepare.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prepare -- python3 $R/tools/time_prepare.py > $R/gpurun_out/prof_prepare.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/prof_prepare -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    n=r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    print("%-60s calls %5s total %9.3f ms avg %9.2f us"%(n[:60],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3))
PY
