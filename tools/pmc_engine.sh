#!/bin/bash
# Runs ON THE GPU BOX: PMC counters per kernel for the bench's batched launches, one counter set per pass (never combined
# with trace domains other than --kernel-trace).  The profiled process spawns nothing: --gen-workers 1 + the scene cache.  Default shape 1 group x 8 scenes ("solo batched": nothing else on the GPU);
#   gpurun -- 'bash tools/pmc_engine.sh r03 1 8'      -> gpurun_out/<tag>_pmc_<pass>.json (per-kernel averages per launch)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}; G=${2:-1}; B=${3:-8}; CACHE=${SG_SCENE_CACHE:-/tmp/sg_scenes}
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters...
  local name=$1; shift
  local out=$R/gpurun_out/${TAG}_pmc_raw_$name
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-files --groups $G --per-group $B --parity-scenes 0 --no-extras --repeats 1 --gen-workers 1 --scene-cache $CACHE > $out.log 2>&1
  python3 - "$out" "$R/gpurun_out/${TAG}_pmc_$name.json" "$G" "$B" <<'PY'
import csv, glob, json, sys, collections, re
src, dst, G, B = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
fs = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*$", "", n)
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"engine": f"{G} groups x {B} scenes per launch", "kernels": {}}
for k, c in agg.items():
    out["kernels"][k] = {"launches": len(next(iter(c.values()))), **{n: sum(v) / len(v) for n, v in c.items()}}
    for n in ("SQ_WAVES", "SQ_INSTS_MFMA"):                    # deterministic per launch: tools/summarise_profiles.py holds min == max == the launch model
        if n in c:
            out["kernels"][k][n + "__min"], out["kernels"][k][n + "__max"] = min(c[n]), max(c[n])
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", kv[1].get("FETCH_SIZE", kv[1].get("WRITE_SIZE", 0)))))[:10]
for k, v in top:
    print("%-44s" % k[:44], {n: round(x) for n, x in v.items()})
PY
  rm -rf $out
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass mfma SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
