#!/bin/bash
# Runs ON THE GPU BOX (round 4, second half): S2X slot-loop variants (packed statistics / packed LeakyReLU product / fp16 halves from v_fma_mix{lo,hi}_f16)
# as micro benchmarks and as whole libraries (bit-identity test + solo-batched stage times), then the layout kernel's block size.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export SG_SCENE_CACHE=/tmp/sg_scenes
for v in 0 1 2 3 4 5 6; do echo "== micro v$v"; timeout 120 build_micro/ec_slots_bench_v$v 2>&1 | grep "S2X" | cut -c1-330; done
for lib in base pkstat lrelupk mixlo mixpk all; do
  echo "== lib $lib"
  SEGGROUP_HIP_LIB=$R/build_micro/lib_ec_$lib.so timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "hand_scheduled or edgeconv" 2>&1 | tail -2
done
for rep in 1 2; do for lib in base pkstat lrelupk mixlo mixpk all; do
  echo "== time $lib"; SEGGROUP_HIP_LIB=$R/build_micro/lib_ec_$lib.so timeout 300 python3 tools/time_engine.py --scene-cache $SG_SCENE_CACHE --tag $lib 2>&1 | tail -1 | cut -c1-500
done; done
for b in 256 128 64 256 128 64; do echo "== layout block $b"; SG_LAYOUT_BLOCK=$b timeout 300 python3 tools/time_engine.py --scene-cache $SG_SCENE_CACHE --tag layout$b 2>&1 | tail -1 | cut -c1-500; done
echo "== single scene host profile"; SG_HOST_PROFILE=1 timeout 300 python3 tools/time_scene.py 150000 1500 10 2>&1 | tail -25
