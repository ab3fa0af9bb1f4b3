"""The C-ABI library loads without a GPU and exports every symbol include/seggroup_hip.h declares;
the ctypes table in seggroup_amd/hip.py covers exactly the same set."""
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "seggroup_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(sg_lib):
    from seggroup_amd import hip
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(sg_lib, n), f"{n} declared in the header but not exported"
    assert sorted(hip.SIGNATURES) == names, "seggroup_amd/hip.py SIGNATURES out of sync with the header"
    assert sg_lib.sg_version() >= 100
    assert sg_lib.sg_device_count() >= 0          # no GPU here: 0, and no crash


def test_library_has_no_runtime_dependency_of_its_own(sg_lib):
    """The .so must bind to the process's HIP runtime (PyTorch's copy), not drag in a second one."""
    import subprocess
    from seggroup_amd import hip
    out = subprocess.run(["readelf", "-d", hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" not in out


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "seggroup_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "cpu_ref" not in txt.replace("oracle/cpu_ref.py only for testing", "").replace(
                    "use oracle/cpu_ref.py for testing", ""), f"{f} references the oracle"
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f"{f} imports the oracle"


def test_missing_device_fails_loudly(sg_lib):
    import pytest
    from seggroup_amd import hip
    if sg_lib.sg_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        hip.require_device()
    import ctypes as C
    w = hip.Weights()
    assert not sg_lib.sg_pipeline_create(10, 1, 1, 10, C.byref(w), None)
    assert b"no HIP device" in sg_lib.sg_last_error() or b"null" in sg_lib.sg_last_error()


def test_scene_size_limit_is_rejected_at_creation(sg_lib):
    """SG_MAX_POINTS (the kNN list keys' 20 index bits): a larger capacity fails when the pipeline / engine is CREATED, not at the first forward"""
    import ctypes as C
    from seggroup_amd import hip
    w = hip.Weights()
    too_many = (1 << 20) + 1
    assert not sg_lib.sg_pipeline_create(too_many, 1, 1, too_many, C.byref(w), None)
    assert b"at most 1048576 points" in sg_lib.sg_last_error()
    assert not sg_lib.sg_engine_create(too_many, 1, 1, too_many, C.byref(w), 1, 1)
    assert b"at most 1048576 points" in sg_lib.sg_last_error()
