#!/bin/bash
# Host-only objects of the library (grouping engine, label / seg.json writers, file parsers, error plumbing) under
# AddressSanitizer + UBSan on the CPU box:  builds build_asan/libseggroup_host_asan.so with g++ and runs the host-engine
# and file-format tests against it (GPU sanitizers are not available on this pool).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
make -C "$R/seggroup_amd/csrc" asan >/dev/null
export SEGGROUP_HIP_HOST_LIB="$R/seggroup_amd/csrc/build_asan/libseggroup_host_asan.so"
export LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)"
# python itself leaks by design; interceptors must not abort on its allocator games
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
cd "$R"
exec python3 -m pytest tests/test_host_engine.py tests/test_hostio.py -x -q -p no:cacheprovider "$@"
