#!/bin/bash
# Runs ON THE GPU BOX: where do the EdgeConv launches spend their cycles -- K scaling + SQ / LDS counters (separate PMC passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
bash tools/prof_edgeconv_k.sh 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  local out=$R/gpurun_out/pmc_ec_$name
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-files --groups 1 --per-group 8 --parity-scenes 0 --no-extras --repeats 1 --gen-workers 1 --scene-cache $SG_SCENE_CACHE > $out.log 2>&1
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
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass c SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
