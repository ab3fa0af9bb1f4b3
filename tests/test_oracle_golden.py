"""Pins oracle/cpu_ref.py to the golden vectors captured from the REAL reference
(tools/capture_reference.py, capture B = contiguous-patched; labels also equal capture A).

Integers (14 label vectors, adjacency lists, cluster ids, metric counts): bit-exact.
Floats: within 1e-5 of the capture, except downstream of a kNN rank-k score tie: torch.topk leaves
the order of equal scores unspecified and the build defines "lower index wins" (oracle docstring of
topk_desc); rows whose neighbour set differs from the capture must be exact ties, and the layers
they feed (BatchNorm batch statistics couple every row) get a 5e-4 band instead.
"""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, make_fixture_scene

FULL = ["tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"]
_cache = {}


def _oracle(golden_index, weight_sets, name, mode):
    from oracle import cpu_ref
    key = (name, mode)
    if key not in _cache:
        _cache[key] = cpu_ref.forward_scene(make_fixture_scene(golden_index, name), weight_sets[mode], mode, keep=True)
    return _cache[key]


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("mode", ["ins_infer", "sem_infer"])
def test_labels_metrics_bit_exact(golden_index, weight_sets, name, mode):
    g = load_golden(name)
    r = _oracle(golden_index, weight_sets, name, mode)
    pre = mode[:3]
    assert golden_index[name][mode]["labels_A_equal_B"], "fixture seed must be reference-stable (capture A == B)"
    assert r["trace"][1:] == golden_index[name][mode]["nclusters"]
    for k, v in r["labels"].items():
        assert np.array_equal(v, g[f"{pre}.label.{k}"]), k
        assert hashlib.sha256(np.ascontiguousarray(v.astype(np.int32)).tobytes()).hexdigest() == golden_index[name][mode]["label_sha"][k]
    for i in range(3):
        assert np.array_equal(r["metrics"][i], g[f"{pre}.metric.{i}"], equal_nan=True)
    assert not r["stalled"]


@pytest.mark.parametrize("name", FULL)
def test_stage_tensors(golden_index, weight_sets, name):
    from oracle import cpu_ref
    g = load_golden(name)
    r = _oracle(golden_index, weight_sets, name, "ins_infer")
    st = r["stages"]
    sc = make_fixture_scene(golden_index, name)
    # island_20k moves a blob to x ~ 20..28 m: the reference's fp32 mean in get_cluster_pointcloud (model.py:422) is ~7e-6 off there,
    # which flips near-tied kNN-10 ranks inside MLP1.  The oracle reproduces torch's fp32 mean bit for bit since round 2 (four
    # interleaved accumulators, cpu_ref.sample_clusters), so this fixture pins floats like the others: samples bit-equal, features and
    # distances within 1e-4 (observed: 2e-6 on MLP1, 1.4e-5 on the GCN outputs).  Round 3 still compared it in a 0.5 band.
    far = name == "island_20k"
    assert np.abs(g["ins.data_1"] - st["samples"]).max() < 3e-6                         # FPS picks + transform
    d = np.abs(g["ins.feat.mlp_1"] - st["feat1"])
    assert d.max() < 1e-5
    assert np.abs(g["ins.dists.0"] - st["d1"]).max() < 1e-4
    for i, a in enumerate((st["adj1"], st["adj2"], st["mlp_2"]["adj"], st["mlp_3"]["adj"])):
        assert np.array_equal(g[f"ins.adj.{i}"].reshape(-1, 2), a), f"adj_{i + 1}"
    for i, root in enumerate((st["root2"], st["mlp_2"]["root"], st["mlp_3"]["root"], st["root5"])):
        assert np.array_equal(g[f"ins.cluster_id.{i}"], root), f"cluster ids after grouping {i}"
    tie_rows = [0, 0]
    if "ins.knn.0" in g.files:
        xyz = sc.data[:, :3]
        for i, nm in enumerate(("mlp_2", "mlp_3")):
            a, b = g[f"ins.knn.{i}"].astype(np.int64), st[nm]["knn"]
            diff = np.nonzero(np.any(np.sort(a, 1) != np.sort(b, 1), axis=1))[0]
            tie_rows[i] = diff.size
            for q in diff:   # must be exact score ties at the k-th rank
                sa = np.sort(cpu_ref.knn_scores(xyz[q][None], xyz[a[q]])[0])
                sb = np.sort(cpu_ref.knn_scores(xyz[q][None], xyz[b[q]])[0])
                assert np.array_equal(sa, sb), f"{nm} row {q}: neighbour sets differ beyond a score tie"
            assert diff.size <= 0.03 * sc.num_points
            same = np.setdiff1d(np.arange(sc.num_points), diff)
            pf = g[f"ins.feat.{nm}"][0].T
            assert np.abs(pf[same] - st[nm]["point_feat"][same]).max() < (1e-5 if diff.size == 0 else 5e-4), nm
    band2 = 1e-5 if tie_rows[0] == 0 and "ins.knn.0" in g.files else 5e-4
    band3 = 1e-5 if sum(tie_rows) == 0 and "ins.knn.0" in g.files else 5e-4
    if far:
        band2 = band3 = 1e-4                                  # no kNN table is stored for this fixture: the float bar of the north star
    assert np.abs(g["ins.feat.gcn_2"] - st["mlp_2"]["gcn"]).max() < band2
    assert np.abs(g["ins.feat.gcn_3"] - st["mlp_3"]["gcn"]).max() < band3
    assert np.abs(g["ins.dists.2"] - st["mlp_2"]["d"]).max() < max(band2, 1e-4)
    assert np.abs(g["ins.dists.4"] - st["mlp_3"]["d"]).max() < max(band3, 1e-4)
    assert np.abs(g["ins.feat5"] - st["feat5"]).max() < band3
    assert np.array_equal(g["ins.adj5"].reshape(-1, 2), st["adj5"])


def test_faithful_mode_equals_vectorised(golden_index, weight_sets):
    """The per-edge Python loops of the reference (faithful=True, timed as the CPU baseline) and the
    vectorised NumPy forms give identical results."""
    from oracle import cpu_ref
    sc = make_fixture_scene(golden_index, "tiny_4k")
    a = _oracle(golden_index, weight_sets, "tiny_4k", "ins_infer")
    b = cpu_ref.forward_scene(sc, weight_sets["ins_infer"], "ins_infer", faithful=True)
    for k in a["labels"]:
        assert np.array_equal(a["labels"][k], b["labels"][k])
    assert cpu_ref.format_label_lines(a["labels"]["final.ins"], True) == cpu_ref.format_label_lines(a["labels"]["final.ins"], False)


def test_reference_quirks():
    """Micro-fixtures of SURVEY.md 8c (probe values taken from the real reference)."""
    from oracle import cpu_ref as O
    pts = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0], [0, 0, 0], [2, 0, 0]], np.float32)
    assert O.fps(pts, 4).tolist() == [4, 0, 1, 0]
    assert O.fps_with_fixup(pts, 4).tolist() == [4, 0, 1, 4]

    class L:
        members = [np.array([5, 6, 7])]
        count = 1
    xyz = np.random.default_rng(0).uniform(size=(8, 3)).astype(np.float32)
    row = O.cluster_knn(xyz, L, 20)[5]
    assert row.tolist() == [5, 6, 7] + [0] * 17                        # n <= k: remaining columns stay 0
    x = np.random.default_rng(1).normal(size=(3, 4)).astype(np.float32)
    assert abs(O.edge_distance(x, np.array([[1, 1]]))[0] - 2e-6) < 1e-12   # pairwise_distance(x, x), D = 4
    # union semantics: veto, label copy through -l1*l2 (instance id 0 survives), stale dead roots
    p = O.Partition(np.array([0, -1, 3, 3]), np.array([7, -1, 2, 2]), np.array([0, 1, 2, 3]))
    assert not p.union(0, 2) and p.root.tolist() == [0, 1, 2, 3]       # both labelled, different -> veto
    assert p.union(1, 0) and p.ins[0] == 0 and p.sem[0] == 7           # -(-1*0) = 0 keeps instance 0
    assert p.union(2, 3) and p.members(3).tolist() == [3, 2]           # id2's members first
    assert not p.union(2, 3)                                           # dead root: no points move ...
    assert p.npts[3] == 3.0                                            # ... but the stale count is added again


def test_group_nearby_stall_is_detected():
    """A <5-point labelled cluster whose only neighbour carries a different label can never merge:
    the reference loops forever (model.py:228-239); the oracle stops and reports it."""
    from oracle import cpu_ref as O
    seg = np.array([0, 0, 0, 1, 1, 1, 1, 1, 1])
    p = O.Partition(np.where(seg == 0, 1, 2), np.where(seg == 0, 5, 6), seg)
    L = O.Layer(p)
    conn, stalled = O.group_nearby(p, np.array([9.0], np.float32), np.array([[0, 1]]), L, 6)
    assert stalled and not conn[0]


@pytest.mark.parametrize("name", ["tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"])
def test_oracle_train_tail_matches_reference_capture(golden_index, weight_sets, name):
    """the oracle's restatement of the train-mode tail (model.py:900-932, Classifier, label-smoothed
    CE) against the capture of the real reference with the same pinned dropout mask (tools/capture_train.py)."""
    import os
    from conftest import GOLDEN, make_fixture_scene
    from oracle import cpu_ref
    g = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    scene = make_fixture_scene(golden_index, name)
    o = cpu_ref.forward_scene(scene, weight_sets["ins_infer"], "ins_infer", keep=True)["stages"]
    Wc = {k[2:]: g[k] for k in g.files if k.startswith("w.classifier.")}
    K = np.unique(o["ins5"]).shape[0]
    t = cpu_ref.train_tail(o["feat5"], o["ins5"], o["sem5"], Wc, cpu_ref.dropout_keep(K))
    assert np.array_equal(g[f"{name}.keep"], cpu_ref.dropout_keep(K))
    assert np.abs(t["feat6"] - g[f"{name}.feat6"]).max() < 1e-4
    assert np.abs(t["logits"] - g[f"{name}.logits"]).max() < 1e-4
    want = g[f"{name}.loss"]
    assert t["loss"][1] == want[0, 1] and abs(t["loss"][0] - want[0, 0]) < 1e-4 * abs(want[0, 0])


@pytest.mark.parametrize("name,tol", [("tiny_4k", 1e-3), ("small_20k", 5e-5)])
def test_oracle_training_step_gradients_match_reference_capture(golden_index, weight_sets, name, tol):
    """the torch-autograd restatement of the differentiable chain (oracle/train_ref.py, float64, on the discrete
    structure the oracle's own forward records) against the gradients the REAL reference leaves on every parameter after
    loss.backward() (tests/golden/train_grads.npz, tools/capture_train.py).  tiny_4k ends with K = 2 instances: BatchNorm1d over
    two rows amplifies the reference's own fp32 rounding (3.5e-4)."""
    import os
    from conftest import GOLDEN, make_fixture_scene
    from oracle import train_ref
    gt = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    gg = np.load(os.path.join(GOLDEN, "train_grads.npz"))
    W = dict(weight_sets["ins_infer"])
    W.update({k[2:]: gt[k] for k in gt.files if k.startswith("w.")})
    r = train_ref.training_step(make_fixture_scene(golden_index, name), W)
    assert abs(r["step_loss"] - float(gg[f"{name}.step_loss"][0])) < 1e-5 * abs(r["step_loss"])
    for k in train_ref.PARAM_KEYS:
        want = gg[f"{name}.grad.{k}"].reshape(-1).astype(np.float64)
        assert np.abs(r["grads"][k].reshape(-1) - want).max() <= tol * np.abs(want).max(), k
    # the batch statistics this chain normalises with reproduce the reference's running buffers after one step
    for k, (m, v, rows) in r["bn"].items():
        want_m, want_v = gg[f"{name}.buf.{k}.running_mean"], gg[f"{name}.buf.{k}.running_var"]
        assert np.abs(0.1 * m - want_m).max() < 1e-5 * max(1.0, np.abs(want_m).max())
        assert np.abs(0.9 + 0.1 * v * rows / (rows - 1) - want_v).max() < 1e-4 * max(1.0, np.abs(want_v).max())


def test_150k_float_stages_against_capture_B(golden_index, weight_sets):
    """BASELINE.json configs[1] at full size, floats: scene_150k.npz holds every 64th point row of the reference's MLP2 / MLP3
    outputs (capture B), its GCN outputs and decision distances, and the rows of its two in-cluster kNN tables that differ from
    the build's defined tie rule (torch.topk leaves the order of equal scores open).  Checked here:
      * the oracle's kNN tables equal the reference's after patching exactly the stored rows (sha256 of the full tables), and
        every patched row is an exact score tie (same score multiset) -- nothing else distinguishes the two;
      * point features of all other sampled rows agree within 1e-4 (observed 2.4e-5);
      * a tie row moves its own point feature by O(1) and, through the cluster maxima and the GCN's neighbour aggregation,
        everything downstream a little: GCN rows and decision distances agree within 1e-4 on > 90 % of the entries (observed
        95 % / 98 % of the rows, 100 % / 93 % / 100 % of the distances) -- the oracle's own values (defined tie rule) are stored
        beside the reference's as the target the HIP path must hit everywhere (tests/test_gpu_scene.py)."""
    from oracle import cpu_ref
    name = "scene_150k"
    e = golden_index[name]
    g = load_golden(name)
    sc = make_fixture_scene(golden_index, name)
    r = cpu_ref.forward_scene(sc, weight_sets["ins_infer"], "ins_infer", keep=True)
    st = r["stages"]
    stride = e["taps_stride"]
    xyz = sc.data[:, :3]
    for nm in ("mlp_2", "mlp_3"):
        tab = st[nm]["knn"].astype(np.int32)
        rows, ref_rows = g[f"ins.tap.knn_tie_rows.{nm}"], g[f"ins.tap.knn_tie_ref.{nm}"]
        assert hashlib.sha256(tab.tobytes()).hexdigest() == e["knn_sha"][nm]["defined_tie_rule"]
        patched = tab.copy()
        patched[rows] = ref_rows
        assert hashlib.sha256(patched.tobytes()).hexdigest() == e["knn_sha"][nm]["reference"], nm
        assert 0 < rows.size == e["knn_sha"][nm]["rows_that_differ"] < 0.02 * tab.shape[0]
        for p, a, b in zip(rows[::7], tab[rows][::7], ref_rows[::7]):             # a sample of the tie rows: equal score multisets
            sa = np.sort(cpu_ref.knn_scores(xyz[p:p + 1], xyz[a])[0])
            sb = np.sort(cpu_ref.knn_scores(xyz[p:p + 1], xyz[b])[0])
            assert np.array_equal(sa, sb), (nm, int(p))
        sampled = np.arange(0, sc.num_points, stride)
        clean = ~np.isin(sampled, rows)
        d = np.abs(g[f"ins.tap.{nm}"] - st[nm]["point_feat"][::stride]).max(axis=1)
        assert d[clean].max() < 1e-4 and clean.sum() > 0.97 * sampled.size, (nm, float(d[clean].max()))
        og = g[f"ins.oracle.gcn_{nm[-1]}"]
        assert np.abs(og - st[nm]["gcn"]).max() < 1e-6                              # the stored oracle target is this oracle's
        dg = np.abs(g[f"ins.tap.gcn_{nm[-1]}"] - st[nm]["gcn"]).max(axis=1)
        assert (dg < 1e-4).mean() > 0.9 and dg.max() < 0.05, (nm, float((dg < 1e-4).mean()), float(dg.max()))
    assert np.abs(g["ins.tap.mlp_1"] - st["feat1"]).max() < 1e-5                    # upstream of every kNN-20: no tie rows yet
    for i, dd in enumerate((st["d1"], st["mlp_2"]["d"], st["mlp_3"]["d"])):
        assert np.abs(g[f"ins.oracle.dists.{i}"] - dd).max() < 1e-6
        dv = np.abs(g[f"ins.tap.dists.{i}"] - dd)
        assert (dv < 1e-4).mean() > (0.999 if i == 0 else 0.9) and dv.max() < 0.05, (i, float((dv < 1e-4).mean()), float(dv.max()))
