#!/bin/bash
# Builds build_micro/lib_ec_<name>.so: the release library with the EdgeConv slot loops regenerated under the given generator knobs
# (environment assignments), e.g.   tools/build_ec_variant.sh pk SG_EC_S2X_PKSTAT=1 SG_EC_LRELU_PK=1
# Only kernels_edgeconv.o is recompiled (in a scratch copy of csrc/); the other objects come from seggroup_amd/csrc/build.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
T=/tmp/sg_ec_variant_$NAME
rm -rf $T && mkdir -p $T && cp $R/seggroup_amd/csrc/*.h $R/seggroup_amd/csrc/kernels_edgeconv.hip $T/
env "$@" SG_EC_OUT=$T/edgeconv_slots_gen.h python3 $R/tools/gen_edgeconv_asm.py --experimental 2>&1 | tail -1 | cut -c1-60
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -I$R/include -I$T -fno-honor-nans -fno-slp-vectorize \
    -c $T/kernels_edgeconv.hip -o $T/kernels_edgeconv.o
OBJS=$(ls $R/seggroup_amd/csrc/build/*.o | grep -v kernels_edgeconv.o)
mkdir -p $R/build_micro && g++ -shared -fPIC -o $R/build_micro/lib_ec_$NAME.so $OBJS $T/kernels_edgeconv.o
echo built build_micro/lib_ec_$NAME.so
