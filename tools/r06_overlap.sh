#!/bin/bash
# Runs ON THE GPU BOX (round 6): does confining launches to a part of the chip buy overlap?
#   gpurun --timeout 1500 -- 'bash tools/r06_overlap.sh'
# 1. where the bits of a HIP CU mask land (tools/micro/cu_mask_probe.hip, built by the caller into build_micro/)
# 2. the bench (40 steps x 2 regions, 8 scenes checked against the single pipeline) per variant of SG_ENGINE_CUMASK (engine.cpp) /
#    SG_EC_WALK (persistent EdgeConv workgroups per CU): scenes/s, EdgeConv / kNN / tail stage times inside the timed region
R=${GRAFT_REPO_ROOT:-/root/repo}
export SG_SCENE_CACHE=/tmp/sg_scenes
cd $R; mkdir -p gpurun_out
OUT=gpurun_out/r06_overlap.txt; : > $OUT
python3 bench.py --generate-only --no-extras --scene-cache $SG_SCENE_CACHE 2>&1 | tail -1
if [ -x build_micro/cu_mask_probe ]; then timeout 60 build_micro/cu_mask_probe | tee -a $OUT; fi
run() {
  name=$1; shift
  env "$@" timeout 300 python3 bench.py --steps ${SG_STEPS:-40} --repeats 2 --warmup 6 --no-cpu-baseline --no-files --no-extras --no-oos --parity-scenes 8 --scene-cache $SG_SCENE_CACHE 2> gpurun_out/r06_overlap_$name.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
except Exception as e:
    print('%-16s FAILED' % '$name'); sys.exit(0)
st=d['roofline']['stage_ms_in_timed_region']; k=d['roofline']['kernels']
heavy=sum(st.get(x,0) for x in ('kernel.l2.edgeconv','kernel.l3.edgeconv','l2.knn','l3.knn'))
print('%-16s %s parity %s | in-region ms/scene: S2X %.3f S1X %.3f knn %.3f+%.3f fps64 %.3f mlp1 %.3f | solo S2X %.4f S1X %.4f knn %.4f' % ('$name', d['repeat_values']['scenes_per_s'], d['parity_check']['ranks_equal'],
      st.get('kernel.l3.edgeconv',0), st.get('kernel.l2.edgeconv',0), st.get('l2.knn',0), st.get('l3.knn',0), st.get('fps64',0), st.get('mlp1',0),
      k.get('k_edgeconv<S2X>',{}).get('ms_per_scene_launch_solo_batched',0), k.get('k_edgeconv<S1X>',{}).get('ms_per_scene_launch_solo_batched',0), k.get('k_cluster_knn_sorted',{}).get('ms_per_scene_launch_solo_batched',0)))
" | tee -a $OUT
}
if [ $# -gt 0 ]; then for v in "$@"; do IFS=, read -r name a b c <<< "$v"; run $name $a $b $c; done; exit 0; fi
run base         SG_X=0
run walk1.75     SG_EC_WALK=1.75
run walk1.5      SG_EC_WALK=1.5
run all224       SG_ENGINE_CUMASK=all:224 SG_EC_WALK=1.75
run rot32        SG_ENGINE_CUMASK=rot:32 SG_EC_WALK=1.75
run rot64        SG_ENGINE_CUMASK=rot:64 SG_EC_WALK=1.5
run heavy224     SG_ENGINE_CUMASK=heavy:224 SG_EC_WALK=1.75
run heavy224knn  SG_ENGINE_CUMASK=heavy:224:knn SG_EC_WALK=1.75
run heavy192     SG_ENGINE_CUMASK=heavy:192 SG_EC_WALK=1.5
run heavy240     SG_ENGINE_CUMASK=heavy:240 SG_EC_WALK=1.875
run halfcu       SG_ENGINE_CUMASK=halfcu SG_EC_WALK=1.0
run halfxcd      SG_ENGINE_CUMASK=halfxcd SG_EC_WALK=1.0
run base2        SG_X=0
