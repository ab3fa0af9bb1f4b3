import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_index():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def weight_sets():
    from seggroup_amd import weights
    return {"ins_infer": weights.load_npz(os.path.join(GOLDEN, "weights_g2.npz")),
            "sem_infer": weights.load_npz(os.path.join(GOLDEN, "weights_g1.npz"))}


@pytest.fixture(scope="session")
def sg_lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from seggroup_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return hip.lib()


_scene_cache = {}


def make_fixture_scene(index, name):
    from seggroup_amd import synthetic
    if name not in _scene_cache:
        e = index[name]
        _scene_cache[name] = synthetic.make_scene(e["n"], e["s"], e["seed"], **e["kw"])
    return _scene_cache[name]


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))
