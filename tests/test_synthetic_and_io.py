"""Synthetic generator determinism (the GPU box must regenerate byte-identical inputs) and the
reference on-disk formats."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import make_fixture_scene


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_generator_reproduces_fixture_inputs(golden_index):
    for name in ("tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"):
        sc = make_fixture_scene(golden_index, name)
        for k, want in golden_index[name]["input_sha"].items():
            assert _sha(getattr(sc, k)) == want, f"{name}.{k}: the generator drifted from the captured fixture inputs"
        first = [np.nonzero(sc.seg == s)[0][0] for s in range(sc.num_segments)]
        assert np.all(np.diff(first) > 0)                       # segment numbers ascend with their first point
        assert np.all(np.diff(sc.adj[:, 0] * sc.num_points + sc.adj[:, 1]) > 0) and np.all(sc.adj[:, 0] < sc.adj[:, 1])


def test_reference_tree_roundtrip(tmp_path, golden_index):
    import torch
    from seggroup_amd import synthetic
    from seggroup_amd.scene import seg_from_lists
    sc = make_fixture_scene(golden_index, "tiny_dup_4k")
    synthetic.write_reference_tree(str(tmp_path), [sc])
    base = tmp_path / "dataset" / "scannet"
    assert (base / "scannetv2_train.txt").read_text() == sc.name + "\n"
    lists = json.load(open(base / "label" / "real" / "resampled" / sc.name / (sc.name + ".seg.json")))
    assert len(lists) == sc.num_points
    assert np.array_equal(seg_from_lists(lists, sc.num_points), sc.seg)
    adj = torch.load(base / "adj" / "mesh" / "resampled" / sc.name / (sc.name + ".adj.pth"))
    assert adj.dtype == torch.int64 and np.array_equal(adj.numpy(), sc.adj)
    from seggroup_amd.data import ScanNet
    ds = ScanNet("manual", root=str(tmp_path))
    data, weak, info = ds[0]
    assert len(ds) == 1 and data.shape == (sc.num_points, 6) and weak.shape == (sc.num_points, 2) and int(info) == 0


def test_weights_state_dict_contract():
    """Reference checkpoint layout: module. prefix, both BN aliases, 4-D conv kernels."""
    import torch
    from seggroup_amd import weights
    from seggroup_amd.model import SegModel
    w = weights.make_weights(3, 2.0, affine_jitter=0.1)
    sd = weights.to_state_dict(w)
    assert "module.mlp_3.conv2.1.weight" in sd and sd["module.mlp_1.conv1.0.weight"].shape == (64, 6, 1, 1)
    back = weights.from_state_dict({"epoch": 6, "state_dict": sd, "optimizer": {}})
    for k in w:
        assert np.array_equal(back[k], w[k])
    net = SegModel(exp_name="x", ins_infer=True, data_root="/nonexistent")
    missing = net.load_state_dict({k[len("module."):]: v for k, v in sd.items()}, strict=False)
    assert all("classifier" in k or "running" in k or "num_batches" in k for k in missing.missing_keys) and not missing.unexpected_keys
    assert sum(p.nelement() for p in net.parameters()) == 147880      # FAQ.md:46 / SURVEY: 147,880 parameters
    got = net.export_weights()
    for k in w:
        assert np.array_equal(got[k], w[k])


def test_driver_loads_checkpoints_strictly(tmp_path):
    """infer.py:121 loads `last.t7` with strict key matching; so does the driver here: a complete reference-shaped
    checkpoint loads, one with a missing or a foreign key is refused instead of silently running on random weights."""
    import torch
    from seggroup_amd import infer, weights
    from seggroup_amd.model import SegModel
    w = weights.make_weights(5, 2.0, affine_jitter=0.1)
    full = weights.to_full_state_dict(w)
    assert len(full) == 54 and all(k.startswith("module.") for k in full)           # 54 entries
    net = SegModel(exp_name="x", ins_infer=True, data_root="/nonexistent")
    p = str(tmp_path / "last.t7")
    torch.save({"epoch": 6, "state_dict": full, "optimizer": {}}, p)
    infer.load_checkpoint(net, p)
    got = net.export_weights()
    for k in w:
        assert np.array_equal(got[k], w[k])
    broken = dict(full)
    del broken["module.gcn_3.fc.weight"]
    torch.save({"state_dict": broken}, p)
    with pytest.raises(RuntimeError, match="gcn_3.fc.weight"):
        infer.load_checkpoint(net, p)
    foreign = dict(full)
    foreign["module.mlp_9.conv1.0.weight"] = full["module.mlp_1.conv1.0.weight"]
    torch.save({"state_dict": foreign}, p)
    with pytest.raises(RuntimeError, match="mlp_9"):
        infer.load_checkpoint(net, p)


def test_synthetic_flag_writes_a_tree_once(tmp_path):
    """`infer.py --synthetic N`: the reference's on-disk layout + a checkpoint to resume, and never over an existing tree."""
    import torch
    from seggroup_amd import infer
    from seggroup_amd.data import ScanNet
    args = infer.build_parser().parse_args(["-n", "exp", "--ins_infer", "--root", str(tmp_path), "--synthetic", "2", "--synthetic-points", "2500",
                                            "--synthetic-segments", "25", "-j", "1"])
    assert infer.write_synthetic_tree(args) == 2
    ds = ScanNet(label_style="manual", root=str(tmp_path))
    data, weak, info = ds[1]
    assert data.shape == (2500, 6) and weak.shape[0] == 2500
    sd = torch.load(tmp_path / "checkpoints" / "exp" / "models" / "last.t7")["state_dict"]
    assert any(k.startswith("module.") or "." in k for k in sd)
    with pytest.raises(SystemExit):
        infer.write_synthetic_tree(args)
