"""What the built library must NOT contain (round 5).

* No packed fp32 arithmetic (v_pk_add / mul / fma_f32) in any kernel: on the MI355X boxes of this pool those instructions lost results once waves
  of other kernels shared the SIMD -- a scene's labels depended on the run.  The Makefile switches the target feature off; the generated
  EdgeConv loops are the plain variants.
* No scratch (spilled registers) in the kernels of the inference path: a spill is HBM traffic per lane and tile.  The trainer's backward
  kernel is the one known exception.

Needs no GPU: the code objects are read out of the .so with llvm-objdump / llvm-readelf.
"""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import scratch_report  # noqa: E402

LLVM_OBJDUMP = os.path.join(scratch_report.LLVM, "llvm-objdump")
pytestmark = pytest.mark.skipif(not os.path.exists(LLVM_OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")

# trainer only (one stream per process), 124-132 B | the general-signature knn(x, k) of functional.py (any C, k <= 128: a per-thread list indexed
# at run time, 1 KB; never launched by the pipeline or the engine)
SCRATCH_ALLOWED = ("k_eb_backward_mfma", "k_knn_general")


@pytest.fixture(scope="module")
def lib_path(sg_lib):
    from seggroup_amd import hip
    return hip.LIB_PATH


def test_the_library_holds_gfx950_kernels(lib_path):
    ks = scratch_report.kernels_of_library(lib_path)
    assert len(ks) >= 100, "too few kernels found: the code objects were not read"
    names = [k[0] for k in ks]
    for needle in ("k_edgeconv_hb", "k_cluster_knn_sorted_b", "k_fps_sample_b", "k_gcn_fc_mfma"):
        assert any(needle in n for n in names), needle


def test_no_kernel_of_the_inference_path_uses_scratch(lib_path):
    bad = [(n, b, s) for n, b, _, s in scratch_report.kernels_of_library(lib_path) if b and not any(a in n for a in SCRATCH_ALLOWED)]
    assert not bad, "kernels with scratch (name, bytes, spilled VGPRs): %r" % bad


def test_no_kernel_uses_packed_fp32_arithmetic(lib_path):
    pk = scratch_report.packed_fp32_of_library(lib_path)
    assert not pk, "packed fp32 instructions per kernel: %r" % pk
