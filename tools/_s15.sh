cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
run() {
  name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d2h_$name -- python3 $R/tools/micro/d2h_path.py > $R/gpurun_out/d2h_$name.log 2>&1)
  f=$(find gpurun_out/d2h_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name: $(grep copies gpurun_out/d2h_$name.log | tail -1)"
  grep -i "copyBuffer\|Name" $f | cut -c1-150
  rm -rf gpurun_out/d2h_$name
}
run default SG_X=0
run blit0 GPU_FORCE_BLIT_COPY_SIZE=0
run blit16 GPU_FORCE_BLIT_COPY_SIZE=16
run sdma1 HSA_ENABLE_SDMA=1
run sdma0 HSA_ENABLE_SDMA=0
