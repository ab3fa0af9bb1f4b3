"""Annotation-derived label producers (seggroup_amd/labels.py) against the capture of the real reference functions
(tests/golden/prep_labels.npz, tools/capture_labels.py): the same synthetic scan and annotations, every file and return value."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

os.environ.setdefault("SEGGROUP_HOST_ONLY", "1")
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    import torch
    import capture_labels as cl
    from oracle import prep_ref
    from seggroup_amd import prepare, synthetic
    fx = cl.FIXTURE
    scan = synthetic.make_raw_scan(fx["w"], fx["h"], fx["seed"], name=fx["name"], cell=fx["cell"])
    ann = synthetic.make_annotations(scan, 11, blocks_per_row=-(-fx["w"] // fx["cell"]))
    td = str(tmp_path_factory.mktemp("labels"))
    scene_path = cl.write_inputs(td, scan, ann)
    prepare.write_ply(os.path.join(scene_path, scan.name + "_vh_clean_2.ply"), scan.xyz, scan.rgb, scan.faces)
    # what the GPU-side producers (covered by tests/test_gpu_prepare.py) leave behind, here from the oracle: map.pth and .seg.txt
    mapper = prep_ref.make_mapper(scan.xyz.shape[0], fx["num_points"], scan.perm)
    os.makedirs(os.path.join(td, "data", "resampled", scan.name))
    torch.save(torch.from_numpy(mapper), os.path.join(td, "data", "resampled", scan.name, scan.name + ".map.pth"))
    raw_lab, _ = prep_ref.segment_lists(scan.seg_indices, mapper)
    os.makedirs(os.path.join(td, "label", "real", "raw", scan.name))
    with open(os.path.join(td, "label", "real", "raw", scan.name, scan.name + ".seg.txt"), "w") as f:
        f.write("".join("%d\n" % v for v in raw_lab))
    return td, scene_path, scan


def test_real_labels_and_pth(tree):
    import torch
    from seggroup_amd import labels
    td, scene_path, scan = tree
    g = np.load(os.path.join(GOLD, "prep_labels.npz"))
    labels.generate_real_labels(scene_path + "/", root=td)
    labels.generate_real_label_pth(scene_path, root=td)
    raw = os.path.join(td, "label", "real", "raw", scan.name)
    assert np.array_equal(np.array(labels.load_labels(os.path.join(raw, scan.name + ".ins.txt"))), g["real.ins"])
    assert np.array_equal(np.array(labels.load_labels(os.path.join(raw, scan.name + ".sem.txt"))), g["real.sem"])
    t = torch.load(os.path.join(raw, scan.name + ".label.pth"))
    assert t.dtype == torch.int64 and np.array_equal(t.numpy(), g["real.label_pth"])
    assert (g["real.ins"] == 0).sum() > 100 and g["real.sem"].max() <= 40


@pytest.mark.parametrize("style,kw", [("manual", {}), ("maxseg", {}), ("maxseg", {"anno_num": 2}), ("rand", {}), ("mainseg", {"main_num": 3})])
def test_weak_labels_every_style(tree, style, kw):
    """the clicks of every label style (the random ones under np.random.seed(1), drawn in the reference's order), the files and the
    returned counts"""
    import torch
    import capture_labels as cl
    from seggroup_amd import labels
    td, scene_path, scan = tree
    g = np.load(os.path.join(GOLD, "prep_labels.npz"))
    labels.generate_real_labels(scene_path, root=td)
    d = cl.style_dir(style, kw)
    np.random.seed(1)
    ret = labels.generate_weak_labels(scene_path, None, label_style=style, manual_label_path=os.path.join(td, "manual"), root=td, **kw)
    assert list(ret) == g[f"{d}.ret"].tolist()
    raw = os.path.join(td, "label", "seg", d, "raw", scan.name)
    assert np.array_equal(np.array(labels.load_labels(os.path.join(raw, scan.name + ".ins.txt"))), g[f"{d}.ins"])
    assert np.array_equal(np.array(labels.load_labels(os.path.join(raw, scan.name + ".sem.txt"))), g[f"{d}.sem"])
    labels.generate_weak_label_pth(scan.name, d, root=td)
    t = torch.load(os.path.join(td, "label", "seg", d, "resampled", scan.name, scan.name + ".label.pth"))
    assert t.dtype == torch.int64 and np.array_equal(t.numpy(), g[f"{d}.label_pth"])


def test_group_adjacency_order_and_scene0217_cut(tmp_path):
    from seggroup_amd import labels
    adj = np.zeros((6, 6))
    for a, b in ((0, 1), (1, 4), (2, 3)):
        adj[a, b] = adj[b, a] = 1
    got = labels.group_adjacency_segs(adj, np.array([0, 1, 2, 3, 4, 5]))
    # the later segment's list absorbs the earlier one's (util.py:260-264): list order and member order are the reference's
    assert got == [[3, 2], [4, 1, 0], [5]]
    p = tmp_path / "scene0217_00.aggregation.json"
    p.write_text(json.dumps({"segGroups": [{"objectId": 30, "label": "a", "segments": [1]}, {"objectId": 31, "label": "a", "segments": [2]},
                                           {"objectId": 32, "label": "a", "segments": [3]}]}))
    ins, sem = labels.load_aggregation(str(p), {"a": 7})
    assert ins == {1: 31} and sem == {1: 7}
