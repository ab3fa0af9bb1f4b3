#!/bin/bash
# Runs ON THE GPU BOX: texture-address / L1 / L2 counters of the batched EdgeConv and kNN launches (separate PMC passes, solo batched)
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1

cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  local out=$R/gpurun_out/pmc_ec_$name
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-files --groups 1 --per-group 8 --parity-scenes 0 --no-extras --repeats 1 --gen-workers 1 --scene-cache $SG_SCENE_CACHE > $out.log 2>&1
  python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "edgeconv_b" in n or "knn_sorted_b" in n or "moments_b" in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print("%-42s" % k[:42], {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
  rm -rf $out
}
# (at most two counters of one block per pass: more "exceeds the capabilities of the hardware" and the profiler hangs; every pass under its own timeout)
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pass tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcc1 TCC_HIT_sum TCC_MISS_sum
pass tcc2 TCC_REQ_sum TCC_TAG_STALL_sum
