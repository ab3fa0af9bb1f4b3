# Runs ON THE GPU BOX: the same command with several builds of the library (SEGGROUP_HIP_LIB), e.g.
#   gpurun -- 'bash tools/ab_libs.sh "python tools/time_prepare.py" build_micro/lib_a.so build_micro/lib_b.so'
CMD=$1; shift
for rep in 1 2; do for lib in "$@"; do echo "== $lib"; SEGGROUP_HIP_LIB=$PWD/$lib timeout 300 $CMD 2>&1 | tail -1 | cut -c1-400; done; done
