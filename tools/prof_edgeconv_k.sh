#!/bin/bash
# Runs ON THE GPU BOX: per-kernel durations of sg_edgeconv_forward for K = 4, 10, 20, 40 (fixed per-tile cost vs per-slot cost)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_eck -- python3 $R/tools/time_edgeconv_k.py > $R/gpurun_out/prof_eck.log 2>&1
f=$(ls -t $(find $R/gpurun_out/prof_eck -name "*kernel_trace.csv") | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
seq=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    if "edgeconv" in n or "moments" in n: seq[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in seq.items():
    # 6 calls per (K, layers) config (1 warm + 5 reps); configs in order K=4,10,20,40
    per=len(v)//4
    print("%-40s"%n[:40], ["%.0f"%(sum(v[i*per+1:(i+1)*per])/(per-1)) for i in range(4)])
PY
