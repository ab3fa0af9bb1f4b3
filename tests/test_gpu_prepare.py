"""SURVEY.md 8f-3 on the GPU: the pre-processing kernels (C ABI: sg_prep_sample_points, sg_nearest_point,
sg_mesh_adjacency, sg_segment_lists) and the reference-named file producers of seggroup_amd/prepare.py against the
captures of the REAL reference (tests/golden/prep_*.npz / digests) -- all integer / byte outputs bit-exact, the fp32
point cloud bit-exact as well (it is a gather plus one correctly rounded expression)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
FULL = ["prep_sub_3k", "prep_rep_1k", "prep_exact_2k"]


def _index():
    return json.load(open(os.path.join(GOLD, "prep_index.json")))


def _scan(name):
    from seggroup_amd import synthetic
    e = _index()[name]
    return synthetic.make_raw_scan(e["w"], e["h"], e["seed"], name=name), e


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _lists_from_csr(pts, off, n):
    lists = [[] for _ in range(n)]
    for g in range(len(off) - 1):
        m = pts[off[g]:off[g + 1]].tolist()
        lists[m[0]] = m
    return lists


@pytest.mark.parametrize("name", FULL)
def test_device_functions_match_reference_capture(name):
    from seggroup_amd import prepare
    scan, e = _scan(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    pcl, unmap, missing = prepare.sample_points(scan.xyz, scan.rgb, g["map"], device="cuda:0")
    assert missing == e["unsampled"]
    assert np.array_equal(pcl.cpu().numpy(), g["pcl"])
    assert np.array_equal(unmap.cpu().numpy(), g["unmap"])                       # tie rows included (lowest index wins)
    raw, res = prepare.mesh_adjacency(scan.faces, unmap, device="cuda:0")
    assert raw.dtype == res.dtype and str(raw.dtype) == "torch.int64"
    assert np.array_equal(raw.cpu().numpy(), g["adj_raw"]) and np.array_equal(res.cpu().numpy(), g["adj_resampled"])
    raw_only, none = prepare.mesh_adjacency(scan.faces, None, num_vertices=e["V"], device="cuda:0")
    assert none is None and np.array_equal(raw_only.cpu().numpy(), g["adj_raw"])
    lab, pts, off = prepare.segment_lists(scan.seg_indices, g["map"], device="cuda:0")
    assert "".join("%d\n" % v for v in lab.cpu().numpy()).encode() == g["seg_txt"].tobytes()
    pts, off = pts.cpu().numpy(), off.cpu().numpy()
    assert off[0] == 0 and off[-1] == e["num_points"] and (np.diff(off) > 0).all()
    assert json.dumps(_lists_from_csr(pts, off, e["num_points"])).encode() == g["seg_json"].tobytes()
    # get_unmapper on its own (the reference's signature: two [*,3] clouds)
    miss = np.nonzero(np.bincount(g["map"], minlength=e["V"]) == 0)[0]
    if miss.size:
        idx = prepare.get_unmapper(scan.xyz[miss], g["pcl"][:, :3], device="cuda:0")
        assert np.array_equal(idx.cpu().numpy(), g["unmap"][miss])


def test_60k_scan_matches_reference_digests():
    from seggroup_amd import prepare
    scan, e = _scan("prep_sub_60k")
    from oracle import prep_ref
    mapper = prep_ref.make_mapper(e["V"], e["num_points"], scan.perm)
    assert _sha(mapper) == e["sha"]["map"]
    pcl, unmap, missing = prepare.sample_points(scan.xyz, scan.rgb, mapper, device="cuda:0")
    assert missing == e["unsampled"] and _sha(pcl.cpu().numpy()) == e["sha"]["pcl"]
    assert _sha(unmap.cpu().numpy()) == e["sha"]["unmap"]                        # 350 tie rows among 21,200 searches
    raw, res = prepare.mesh_adjacency(scan.faces, unmap, device="cuda:0")
    assert _sha(raw.cpu().numpy()) == e["sha"]["adj_raw"] and _sha(res.cpu().numpy()) == e["sha"]["adj_resampled"]
    lab, pts, off = prepare.segment_lists(scan.seg_indices, mapper, device="cuda:0")
    assert hashlib.sha256("".join("%d\n" % v for v in lab.cpu().numpy()).encode()).hexdigest() == e["sha"]["seg_txt"]
    text = json.dumps(_lists_from_csr(pts.cpu().numpy(), off.cpu().numpy(), e["num_points"]))
    assert hashlib.sha256(text.encode()).hexdigest() == e["sha"]["seg_json"]


@pytest.mark.parametrize("name", ["prep_sub_3k", "prep_rep_1k"])
def test_reference_named_functions_write_the_reference_files(tmp_path, name):
    """generate_pointcloud_pth / generate_seg_labels_and_ds_set / generate_mesh_adjcency_pth called like
    prepare_data.py:36-71 does, on a scan directory holding `<s>_vh_clean_2.ply` and `<s>_vh_clean_2.0.010000.segs.json`."""
    import torch
    from seggroup_amd import prepare
    scan, e = _scan(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    root = str(tmp_path)
    sp = os.path.join(root, "scans", scan.name)
    os.makedirs(sp)
    prepare.write_ply(os.path.join(sp, scan.name + "_vh_clean_2.ply"), scan.xyz, scan.rgb, scan.faces)
    with open(os.path.join(sp, scan.name + "_vh_clean_2.0.010000.segs.json"), "w") as f:
        json.dump({"segIndices": scan.seg_indices.tolist()}, f)
    prepare.prepare_scene(sp + "/", 5, e["num_points"], root=root, perm=scan.perm, device="cuda:0")
    d = os.path.join(root, "data", "resampled", scan.name)
    ld = lambda p: torch.load(p)
    pcl, mp, un, info = (ld(os.path.join(d, scan.name + s)) for s in (".pcl.pth", ".map.pth", ".unmap.pth", ".info.pth"))
    assert pcl.dtype == torch.float32 and mp.dtype == un.dtype == info.dtype == torch.int64 and info.tolist() == [5]
    assert np.array_equal(pcl.numpy(), g["pcl"]) and np.array_equal(mp.numpy(), g["map"]) and np.array_equal(un.numpy(), g["unmap"])
    a1 = ld(os.path.join(root, "adj", "mesh", "raw", scan.name, scan.name + ".adj.pth"))
    a2 = ld(os.path.join(root, "adj", "mesh", "resampled", scan.name, scan.name + ".adj.pth"))
    assert a1.dtype == a2.dtype == torch.int64 and np.array_equal(a1.numpy(), g["adj_raw"]) and np.array_equal(a2.numpy(), g["adj_resampled"])
    assert open(os.path.join(root, "label", "real", "raw", scan.name, scan.name + ".seg.txt"), "rb").read() == g["seg_txt"].tobytes()
    assert open(os.path.join(root, "label", "real", "resampled", scan.name, scan.name + ".seg.json"), "rb").read() == g["seg_json"].tobytes()


def test_prepared_tree_feeds_the_hot_path(tmp_path, weight_sets):
    """raw scan -> prepare_scene -> (plus label files) -> SegModel's loader -> forward: the labels equal the oracle's
    on the same prepared inputs.  The mesh adjacency carries (a, a) rows and the unmapper is not the identity."""
    import torch
    from oracle import cpu_ref
    from seggroup_amd import hip, prepare, synthetic
    from seggroup_amd.model import SegModel
    from seggroup_amd.scene import DeviceScene, seg_from_lists
    scan = synthetic.make_raw_scan(56, 49, 21, name="scene0021_00", dup_frac=0.02)
    root, n = str(tmp_path), 2000
    sp = os.path.join(root, "scans", scan.name)
    os.makedirs(sp)
    prepare.write_ply(os.path.join(sp, scan.name + "_vh_clean_2.ply"), scan.xyz, scan.rgb, scan.faces)
    with open(os.path.join(sp, scan.name + "_vh_clean_2.0.010000.segs.json"), "w") as f:
        json.dump({"segIndices": scan.seg_indices.tolist()}, f)
    base = os.path.join(root, "dataset", "scannet")
    prepare.prepare_scene(sp, 0, n, root=base, perm=scan.perm, device="cuda:0")
    lists = json.load(open(os.path.join(base, "label", "real", "resampled", scan.name, scan.name + ".seg.json")))
    seg = seg_from_lists(lists, n)
    s = int(seg.max()) + 1
    # annotation-derived files (out of scope of prepare.py): a few labelled segments, ground truth per raw vertex
    weak = np.full((n, 2), -1, np.int64)
    for k, sg in enumerate(range(0, s, max(s // 9, 1))):
        weak[seg == sg] = (k % 5 + 1, k)
    unmap = torch.load(os.path.join(base, "data", "resampled", scan.name, scan.name + ".unmap.pth")).numpy()
    gt = np.stack([np.maximum(weak[unmap, 0], 0) + 1, np.maximum(weak[unmap, 1], 0) + 1], 1).astype(np.int64)
    for sub, arr in ((("label", "seg", "manual", "resampled"), weak), (("label", "real", "raw"), gt)):
        dd = os.path.join(base, *sub, scan.name)
        os.makedirs(dd, exist_ok=True)
        torch.save(torch.from_numpy(arr), os.path.join(dd, scan.name + ".label.pth"))
    ds = DeviceScene.from_reference_tree(scan.name, root=root, device="cuda:0")
    adj = torch.load(os.path.join(base, "adj", "mesh", "resampled", scan.name, scan.name + ".adj.pth")).numpy()
    assert (adj[:, 0] == adj[:, 1]).any() and not np.array_equal(unmap, np.arange(unmap.shape[0]))
    net = SegModel(exp_name="t", ins_infer=True, data_root=root)
    net.load_weights(weight_sets["ins_infer"])
    net.epoch = "ins_infer"
    res = net.forward_scene(ds, write=False)
    data = torch.load(os.path.join(base, "data", "resampled", scan.name, scan.name + ".pcl.pth")).numpy()
    ref = cpu_ref.forward_scene(synthetic.Scene(scan.name, data, weak, seg, adj, unmap, gt), weight_sets["ins_infer"], "ins_infer")
    assert res.trace == ref["trace"]
    for i in range(14):
        assert np.array_equal(res.labels[i], ref["labels"][hip.LABEL_NAMES[i]].astype(np.int32)), hip.LABEL_NAMES[i]


def test_raw_scan_directory_to_training_step(tmp_path, weight_sets):
    """A raw scan directory as ScanNet ships it (mesh, segs.json, aggregation.json, label map, click file) -> prepare_scene with a
    label style -> every input file of the reference's tree -> the inference forward equals the oracle on those files, and one
    training step runs on them."""
    import torch
    from oracle import cpu_ref
    from seggroup_amd import hip, prepare, synthetic, train, trainer as T
    from seggroup_amd.model import SegModel
    from seggroup_amd.scene import DeviceScene, seg_from_lists
    scan = synthetic.make_raw_scan(64, 48, 21, name="scene0021_00", cell=8)
    ann = synthetic.make_annotations(scan, 11, blocks_per_row=8)
    root, n = str(tmp_path), 2500
    base = os.path.join(root, "dataset", "scannet")
    sp = os.path.join(base, "scans", scan.name)
    os.makedirs(sp)
    prepare.write_ply(os.path.join(sp, scan.name + "_vh_clean_2.ply"), scan.xyz, scan.rgb, scan.faces)
    json.dump({"segIndices": scan.seg_indices.tolist()}, open(os.path.join(sp, scan.name + "_vh_clean_2.0.010000.segs.json"), "w"))
    json.dump(ann["aggregation"], open(os.path.join(sp, scan.name + ".aggregation.json"), "w"))
    open(os.path.join(base, "scannetv2-labels.combined.tsv"), "w").write(ann["tsv"])
    os.makedirs(os.path.join(root, "clicks"))
    json.dump(ann["manual"], open(os.path.join(root, "clicks", scan.name + ".json"), "w"))
    open(os.path.join(base, "scannetv2_train.txt"), "w").write(scan.name + "\n")
    prepare.prepare_scene(sp, 0, n, root=base, perm=scan.perm, device="cuda:0", label_style="manual", manual_label_path=os.path.join(root, "clicks"))
    ds = DeviceScene.from_reference_tree(scan.name, root=root, device="cuda:0")
    ld = lambda *q: torch.load(os.path.join(base, *q)).numpy()
    weak = ld("label", "seg", "manual", "resampled", scan.name, scan.name + ".label.pth")
    gt = ld("label", "real", "raw", scan.name, scan.name + ".label.pth")
    assert weak.shape == (n, 2) and (weak[:, 1] >= 0).sum() > 200 and weak.min() == -1 and gt.shape == (scan.xyz.shape[0], 2)
    lists = json.load(open(os.path.join(base, "label", "real", "resampled", scan.name, scan.name + ".seg.json")))
    seg = seg_from_lists(lists, n)
    sc = synthetic.Scene(scan.name, ld("data", "resampled", scan.name, scan.name + ".pcl.pth"), weak, seg,
                         ld("adj", "mesh", "resampled", scan.name, scan.name + ".adj.pth"), ld("data", "resampled", scan.name, scan.name + ".unmap.pth"), gt)
    net = SegModel(exp_name="t", ins_infer=True, data_root=root)
    net.load_weights(weight_sets["ins_infer"])
    net.epoch = "ins_infer"
    res = net.forward_scene(ds, write=False)
    ref = cpu_ref.forward_scene(sc, weight_sets["ins_infer"], "ins_infer")
    assert res.trace == ref["trace"]
    for i in range(14):
        assert np.array_equal(res.labels[i], ref["labels"][hip.LABEL_NAMES[i]].astype(np.int32)), hip.LABEL_NAMES[i]
    st = train.initial_state(1)
    tr = T.Trainer(st, (ds.N, ds.S, ds.E0, ds.V), device="cuda:0")
    loss, _, _ = tr.step(ds)
    assert loss[0, 1] >= 2 and np.isfinite(loss[0, 0]) and torch.isfinite(tr.params).all()
    tr.close()


def test_full_size_scan_properties():
    """ScanNet-sized scan (V = 245k > num_points = 150k: 95k nearest-point searches against 150k samples) through
    size-independent properties: every sampled vertex unmaps to a copy of itself, every unsampled one to a sample no
    farther than a random probe set, the adjacency is strictly sorted and equals the host construction, the segment CSR
    partitions the cloud."""
    import torch
    from seggroup_amd import prepare, synthetic
    scan = synthetic.make_raw_scan(600, 400, 31, dup_frac=0.02)
    v, n = scan.xyz.shape[0], 150000
    from oracle import prep_ref
    mapper = prep_ref.make_mapper(v, n, scan.perm)
    pcl, unmap, missing = prepare.sample_points(scan.xyz, scan.rgb, mapper, device="cuda:0")
    pcl, unmap = pcl.cpu().numpy(), unmap.cpu().numpy()
    sampled = np.zeros(v, bool); sampled[mapper] = True
    assert missing == int((~sampled).sum()) and (unmap >= 0).all() and (unmap < n).all()
    assert np.array_equal(mapper[unmap[sampled]], np.nonzero(sampled)[0])
    assert np.array_equal(pcl[:, :3], scan.xyz[mapper])
    miss = np.nonzero(~sampled)[0][::97]
    # exactly the reference's choice on a subset (~1000 rows x 150k candidates through the oracle) ...
    assert np.array_equal(unmap[miss], prep_ref.get_unmapper(scan.xyz[miss], pcl[:, :3]))
    # ... which is the nearest sample up to the resolution of the expanded fp32 form -|x|^2 + 2xy - |y|^2 (|x|^2 ~ 600 here)
    d_sel = ((scan.xyz[miss].astype(np.float64) - pcl[unmap[miss], :3]) ** 2).sum(1)
    probe = pcl[::53, :3].astype(np.float64)
    d_probe = ((scan.xyz[miss][:, None, :].astype(np.float64) - probe[None]) ** 2).sum(2).min(1)
    assert (d_sel <= d_probe + 16 * 2.0 ** -24 * 2 * float((pcl[:, :3].astype(np.float64) ** 2).sum(1).max())).all()
    raw, res = prepare.mesh_adjacency(scan.faces, unmap, device="cuda:0")
    for a in (raw.cpu().numpy(), res.cpu().numpy()):
        key = a[:, 0] * (1 << 32) + a[:, 1]
        assert (np.diff(key) > 0).all() and (a[:, 0] <= a[:, 1]).all()
    want_raw, want_res = prep_ref.get_adj_from_mesh(scan.faces, unmap)
    assert np.array_equal(raw.cpu().numpy(), want_raw) and np.array_equal(res.cpu().numpy(), want_res)
    lab, pts, off = prepare.segment_lists(scan.seg_indices, mapper, device="cuda:0")
    lab, pts, off = lab.cpu().numpy(), pts.cpu().numpy(), off.cpu().numpy()
    assert np.array_equal(np.sort(pts), np.arange(n)) and off[0] == 0 and off[-1] == n
    lab_s = lab[mapper]
    for g_ in (0, len(off) // 2, len(off) - 2):
        m = pts[off[g_]:off[g_ + 1]]
        assert (np.diff(m) > 0).all() and len(set(lab_s[m].tolist())) == 1 and int((lab_s == lab_s[m[0]]).sum()) == m.size
    assert np.array_equal(lab, np.searchsorted(np.unique(scan.seg_indices), scan.seg_indices))


def test_pointcloud_adjacency_matches_reference_and_oracle():
    """prepare.get_adj_from_pointcloud (sg_pointcloud_adjacency) against the capture of the real reference outside the tie points, and
    against the oracle everywhere (both rank equal scores by ascending index), also on an 8k-point cloud and for k = 5 / 20."""
    import torch
    from oracle import prep_ref
    from seggroup_amd import prepare
    g = np.load(os.path.join(GOLD, "prep_sub_3k.npz"))
    ref = np.load(os.path.join(GOLD, "prep_pointcloud_adj.npz"))
    got = prepare.get_adj_from_pointcloud(torch.from_numpy(g["pcl"]), k=10)
    assert got.dtype == torch.int64 and not got.is_cuda
    got = got.numpy()
    adj, tie = prep_ref.get_adj_from_pointcloud(g["pcl"], k=10)
    assert np.array_equal(got, adj)
    keep = lambda e: e[~(tie[e[:, 0]] | tie[e[:, 1]])]
    assert np.array_equal(keep(got), keep(ref["adj"].astype(np.int64)))
    rng = np.random.default_rng(12)
    cloud = (rng.random((8000, 6)) * np.array([8, 6, 3, 1, 1, 1])).astype(np.float32)
    for k in (5, 10, 20):
        want, _ = prep_ref.get_adj_from_pointcloud(cloud, k=k)
        have = prepare.get_adj_from_pointcloud(cloud, k=k).numpy()
        assert np.array_equal(have, want), k
    with pytest.raises(Exception):
        prepare.get_adj_from_pointcloud(cloud, k=7)
