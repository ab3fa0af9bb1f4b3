"""SURVEY.md 8f-3 on the CPU: the pre-processing oracle (oracle/prep_ref.py) against outputs of the REAL reference
functions (tests/golden/prep_*.npz, written by tools/capture_prepare.py), and the host-only pieces of the product
(PLY reader, `.seg.json` writer)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
FULL = ["prep_sub_3k", "prep_rep_1k", "prep_exact_2k"]


def _index():
    return json.load(open(os.path.join(GOLD, "prep_index.json")))


def _scan(name):
    from seggroup_amd import synthetic
    e = _index()[name]
    return synthetic.make_raw_scan(e["w"], e["h"], e["seed"], name=name), e


@pytest.mark.parametrize("name", FULL)
def test_oracle_reproduces_the_reference_outputs(name):
    from oracle import prep_ref
    scan, e = _scan(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    assert scan.xyz.shape[0] == e["V"] and scan.faces.shape[0] == e["F"]
    pcl, mapper, unmap = prep_ref.sample_points(scan.xyz, scan.rgb, e["num_points"], scan.perm)
    assert pcl.dtype == np.float32 and np.array_equal(pcl, g["pcl"])
    assert np.array_equal(mapper, g["map"]) and np.array_equal(unmap, g["unmap"])
    assert (unmap >= 0).all() and int((np.bincount(mapper, minlength=e["V"]) == 0).sum()) == e["unsampled"]
    raw, res = prep_ref.get_adj_from_mesh(scan.faces, unmap)
    assert np.array_equal(raw, g["adj_raw"]) and np.array_equal(res, g["adj_resampled"])
    lab, lists = prep_ref.segment_lists(scan.seg_indices, mapper)
    assert "".join("%d\n" % v for v in lab).encode() == g["seg_txt"].tobytes()
    assert prep_ref.seg_json_text(lists).encode() == g["seg_json"].tobytes()
    # edge cases the fixtures were built for
    assert (g["adj_raw"][:, 0] < g["adj_raw"][:, 1]).all()                       # zero-length edges dropped (util.py:783)
    if e["unsampled"]:
        assert (g["adj_resampled"][:, 0] == g["adj_resampled"][:, 1]).any()      # collapsed only after unmapping: kept
        assert len(e["tie_rows"]) > 0                                            # coincident vertices: exact score ties


def test_oracle_reproduces_the_60k_digests():
    from oracle import prep_ref
    scan, e = _scan("prep_sub_60k")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    pcl, mapper, unmap = prep_ref.sample_points(scan.xyz, scan.rgb, e["num_points"], scan.perm)
    assert sha(pcl) == e["sha"]["pcl"] and sha(mapper) == e["sha"]["map"] and sha(unmap) == e["sha"]["unmap"]
    raw, res = prep_ref.get_adj_from_mesh(scan.faces, unmap)
    assert sha(raw) == e["sha"]["adj_raw"] and sha(res) == e["sha"]["adj_resampled"]
    lab, lists = prep_ref.segment_lists(scan.seg_indices, mapper)
    assert hashlib.sha256("".join("%d\n" % v for v in lab).encode()).hexdigest() == e["sha"]["seg_txt"]
    assert hashlib.sha256(prep_ref.seg_json_text(lists).encode()).hexdigest() == e["sha"]["seg_json"]
    assert all(e["oracle_equals_reference"].values()) and len(e["tie_rows"]) > 100


def test_ply_roundtrip_and_mesh_arrays(tmp_path):
    from seggroup_amd import prepare
    scan, _ = _scan("prep_rep_1k")
    p = str(tmp_path / "m.ply")
    prepare.write_ply(p, scan.xyz, scan.rgb, scan.faces)
    mesh = prepare.read_ply(p)
    assert mesh["vertex"].count == scan.xyz.shape[0] and mesh["face"].count == scan.faces.shape[0]
    xyz, rgb, faces = prepare.mesh_arrays(mesh)
    assert np.array_equal(xyz, scan.xyz) and np.array_equal(rgb, scan.rgb) and np.array_equal(faces, scan.faces)
    assert mesh["vertex"]["alpha"].min() == 255
    # plyfile-style access: a list of per-face index arrays
    class E:
        def __init__(self, c, n): self._c, self.count = c, n
        def __getitem__(self, k): return self._c[k]
    fake = {"vertex": mesh["vertex"], "face": E({"vertex_indices": [f for f in scan.faces]}, scan.faces.shape[0])}
    assert np.array_equal(prepare.mesh_arrays(fake)[2], scan.faces)
    (tmp_path / "bad.ply").write_bytes(b"ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        prepare.read_ply(str(tmp_path / "bad.ply"))


@pytest.mark.parametrize("name", FULL)
def test_seg_json_writer_is_byte_identical_to_json_dump(tmp_path, sg_lib, name):
    """sg_write_seg_json (host only) from the CSR in ANY group order == the reference's json.dump output."""
    from oracle import prep_ref
    scan, e = _scan(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    _, lists = prep_ref.segment_lists(scan.seg_indices, g["map"])
    groups = [l for l in lists if l]
    groups = groups[::-1]                                                        # not the natural order
    pts = np.concatenate([np.asarray(l, np.int32) for l in groups])
    off = np.concatenate([[0], np.cumsum([len(l) for l in groups])]).astype(np.int32)
    p = str(tmp_path / "s.json")
    assert sg_lib.sg_write_seg_json(p.encode(), pts.ctypes.data, off.ctypes.data, len(groups), len(lists)) == 0
    assert open(p, "rb").read() == g["seg_json"].tobytes()
    assert json.load(open(p)) == lists
    bad_off = off.copy(); bad_off[1] = 0
    assert sg_lib.sg_write_seg_json(p.encode(), pts.ctypes.data, bad_off.ctypes.data, len(groups), len(lists)) < 0


def test_oracle_pointcloud_adjacency_matches_reference_capture():
    """get_adj_from_pointcloud (util.py:814-834; optional in the reference) on the sampled cloud of prep_sub_3k, k = 10: the oracle's
    rows equal the real reference's wherever no two of a point's leading scores are equal (torch.topk leaves that order open; the
    capture lists those points)."""
    from oracle import prep_ref
    g = np.load(os.path.join(GOLD, "prep_sub_3k.npz"))
    ref = np.load(os.path.join(GOLD, "prep_pointcloud_adj.npz"))
    adj, tie = prep_ref.get_adj_from_pointcloud(g["pcl"], k=10)
    assert np.array_equal(np.nonzero(tie)[0], ref["tie_points"])
    keep = lambda e: e[~(tie[e[:, 0]] | tie[e[:, 1]])]
    want = ref["adj"].astype(np.int64)
    assert np.array_equal(keep(adj), keep(want)) and keep(adj).shape[0] > 8000
    assert np.all(adj[:, 0] <= adj[:, 1]) and np.array_equal(adj, np.unique(adj, axis=0))   # (i, i) rows: a duplicate outranked the point itself
