#!/bin/bash
# Builds a PROFILING copy of the library (make PROFILE=1: kNN work counters, EdgeConv phase stamps) as build_micro/libseggroup_hip_prof.so
# without touching the release build.  Use: SEGGROUP_HIP_LIB=$PWD/build_micro/libseggroup_hip_prof.so python tools/ec_phases.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/sg_prof_tree
mkdir -p $T/seggroup_amd && rm -rf $T/seggroup_amd/csrc $T/include
cp -r $R/seggroup_amd/csrc $T/seggroup_amd/ && rm -rf $T/seggroup_amd/csrc/build && cp -r $R/include $T/
make -C $T/seggroup_amd/csrc PROFILE=1 -j8 2>&1 | grep -v "^make\|hipcc" | tail -3
mkdir -p $R/build_micro && cp $T/seggroup_amd/libseggroup_hip.so $R/build_micro/libseggroup_hip_prof.so
