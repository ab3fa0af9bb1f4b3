cd $GRAFT_REPO_ROOT
export SG_SCENE_CACHE=/tmp/sg_scenes
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
R=$GRAFT_REPO_ROOT
for shape in "10 8" "1 8" "16 8"; do
  set -- $shape
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $R/gpurun_out/busy_$1x$2 -- python3 $R/bench.py --profile --steps 12 --warmup 3 --repeats 1 --no-cpu-baseline --no-files --groups $1 --per-group $2 --parity-scenes 0 --no-extras --gen-workers 1 --scene-cache $SG_SCENE_CACHE > $R/gpurun_out/busy_$1x$2.log 2>&1)
  echo "== engine $1 x $2"
  python3 tools/gpu_busy_from_trace.py gpurun_out/busy_$1x$2
  grep -o '"value": [0-9.]*' gpurun_out/busy_$1x$2.log | head -1
  f=$(find gpurun_out/busy_$1x$2 -name "*marker_api_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
want=["P0.sync","L2.sync","L3.sync","END.sync","P0.contract","P0.sort_boxes+fps64","P0.mlp1","P0.edge_distance","L2.layout","L2.knn","L2.edgeconv","L2.gcn+edge_distance","L3.layout","L3.knn","L3.edgeconv","L3.gcn+edge_distance","END.export+evaluate","P0.describe","L2.describe","L3.describe","P0.host_grouping","L2.host_grouping","L3.host_grouping","END.final_clustering"]
by={r["Name"]:r for r in rows}
print("   per group super-step, ms: "+", ".join("%s %.2f"%(k, float(by[k]["AverageNs"])/1e6) for k in want if k in by))
PY
  rm -rf gpurun_out/busy_$1x$2
done
