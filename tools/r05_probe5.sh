#!/bin/bash
# EdgeConv walk A/B: HBM traffic (FETCH_SIZE / WRITE_SIZE per launch) and the loaded bench, SG_EC_WALK_MODE = 1 (contiguous, XCD-grouped, carried maxima) vs 0
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R; mkdir -p gpurun_out
timeout 600 python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>/dev/null
for M in 1 0; do
  export SG_EC_WALK_MODE=$M
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/p5_raw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-files --groups 1 --per-group 8 --parity-scenes 1 --no-extras --repeats 1 --gen-workers 1 --scene-cache $SG_SCENE_CACHE > $R/gpurun_out/p5_raw.log 2>&1)
    python3 - $R/gpurun_out/p5_raw $M $C <<'PY'
import csv, glob, sys, collections, re
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if "edgeconv_hb" in n or "edge_moments_b" in n: agg[n].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print("walk", sys.argv[2], sys.argv[3], k, "per scene-launch: %.2f (raw counter units/8)" % (sum(v) / len(v) / 8), "launches", len(v))
PY
    rm -rf $R/gpurun_out/p5_raw
  done
  timeout 600 python3 bench.py --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-files --no-extras --scene-cache $SG_SCENE_CACHE > gpurun_out/p5_bench_$M.json 2> gpurun_out/p5_bench_$M.err
  python3 -c "import json;d=json.load(open('gpurun_out/p5_bench_$M.json'));print('walk $M bench',d['repeat_values']['scenes_per_s'])"
done
