#!/bin/bash
# Runs ON THE GPU BOX (round 6, DESIGN.md section 5 "Round 6"): the measured variant of "BatchNorm 2's statistics from second moments of h1".
# Libraries built beforehand in the build container (tools/build_ec_variant.sh, the generator's --experimental knobs; results of the variants
# are garbage, only their time counts):
#   base  the release slot loops                         q4 / q8  as q plus 4 / 8 more MFMAs per slot into the freed registers, A fragments from LDS
#   q     without sums and sums of squares (maxima only: 64 of a slot's 244 VALU gone -- the most the lever could ever save)
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R
python3 bench.py --generate-only --no-extras --batch 16 --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
for rep in 1 2; do
for v in base q q4 q8; do
  SEGGROUP_HIP_LIB=$R/build_micro/lib_ec_$v.so python3 tools/time_engine.py --tag $v --rounds 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-5s S2X %6.1f us per scene   S1X %5.1f   (solo batched, one group of 8)' % (d['tag'], d['us_per_scene']['kernel.l3.edgeconv'], d['us_per_scene']['kernel.l2.edgeconv']))"
done; done | tee gpurun_out/r06_ec_lever.txt
