#!/bin/bash
# Runs ON THE GPU BOX: bench throughput for several engine shapes (groups x scenes per batched launch)
R=${GRAFT_REPO_ROOT:-/root/repo}
for gb in "$@"; do
  g=${gb%x*}; b=${gb#*x}
  SG_ENGINE_PROFILE=1 timeout 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-files --no-extras --groups $g --per-group $b > /tmp/o.json 2> /tmp/o.err
  python3 - <<PY
import json
try:
    d=json.load(open('/tmp/o.json'))
    print("$gb", d['value'], 'scenes/s', d.get('engine_profile'), {k:round(v,3) for k,v in d['roofline']['stage_ms'].items() if k in ('l2.knn','l3.knn','l2.edgeconv','l3.edgeconv','evaluate','mlp1','fps64')})
except Exception as e:
    print("$gb", 'failed', e)
PY
  grep "engine profile" /tmp/o.err | tail -1
done
