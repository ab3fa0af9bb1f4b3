#!/bin/bash
# Builds build_micro/libsg_knn_r5_nan_bound.so: the library with the round-5 fault of the list-form kNN put back (-DSG_KNN_R5_NAN_BOUND: a published
# bound below -inf's score bits decodes to a NaN).  The kNN gates must FAIL on it:
#   SEGGROUP_HIP_LIB=$PWD/build_micro/libsg_knn_r5_nan_bound.so python -m pytest tests/test_gpu_ops.py -k slices_short        (on the GPU box)
# which is how a change to those tests is shown to still catch what they are there for.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/seggroup_amd/csrc
mkdir -p $R/build_micro/knn_gate
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-function -I../../include -I."
/opt/rocm/bin/hipcc $FLAGS -DSG_KNN_R5_NAN_BOUND -c kernels_knn_sorted.hip -o $R/build_micro/knn_gate/kernels_knn_sorted.o 2> >(grep -v "is not a recognized feature" >&2)
OBJS=$(ls build/*.o | grep -v kernels_knn_sorted.o)
g++ -shared -fPIC -o $R/build_micro/libsg_knn_r5_nan_bound.so $OBJS $R/build_micro/knn_gate/kernels_knn_sorted.o
ls -la $R/build_micro/libsg_knn_r5_nan_bound.so
