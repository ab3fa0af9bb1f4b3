"""The host-only objects of the library under AddressSanitizer + UBSan (sanitizers run on the CPU
build; the GPU pool offers none).  Builds `make asan` with g++ and re-runs the host-engine and file-format tests against
that library in a child process with the sanitizer runtimes preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_engine_and_parsers_are_clean_under_asan_ubsan():
    if os.environ.get("SEGGROUP_HIP_HOST_LIB"):
        pytest.skip("already running inside the sanitizer child")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "run_asan_host_tests.sh")], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
