"""Scene-level parity on the GPU: SegModel.forward_scene (C ABI: sg_pipeline_forward) vs the golden
vectors captured from the real reference and vs the NumPy oracle.  Integer label vectors and metric
counts must be bit-exact; float stage taps within 1e-4 (north_star tolerance)."""
import hashlib

import os

import numpy as np
import pytest

from conftest import load_golden, make_fixture_scene

pytestmark = pytest.mark.gpu
FLOAT_TOL = 1e-4


def _run(scene, w, mode_name, debug=False, knn_variant=None):
    import ctypes as C
    import torch
    from seggroup_amd import hip
    from seggroup_amd.model import SegModel
    from seggroup_amd.scene import DeviceScene

    net = SegModel(exp_name="t", sem_infer=(mode_name == "sem_infer"), ins_infer=(mode_name == "ins_infer"))
    net.load_weights(w)
    net.epoch = mode_name
    ds = DeviceScene.from_synthetic(scene, device="cuda:0")
    pipe = net.pipeline_for(ds)
    if knn_variant is not None:                     # per-pipeline setting: the library keeps no process-wide knob
        pipe.lib.sg_pipeline_set_knn_variant(pipe.handle, knn_variant)
    taps = {}
    dbg = None
    if debug:
        N, S = ds.N, ds.S
        dev = "cuda:0"
        taps["samples1"] = torch.zeros(S, 64, 6, device=dev)
        taps["feat1"] = torch.zeros(S, 128, device=dev)
        taps["pf"] = [torch.zeros(N, 64, device=dev) for _ in range(2)]
        taps["knn"] = [torch.zeros(N, 20, dtype=torch.int32, device=dev) for _ in range(2)]
        taps["members"] = [torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(2)]
        taps["gcn"] = [np.zeros((S, 192), np.float32), np.zeros((S, 256), np.float32)]
        taps["dist"] = [np.zeros(max(ds.E0, 1), np.float32) for _ in range(3)]
        taps["adj"] = [np.zeros((max(ds.E0, 1), 2), np.int32) for _ in range(4)]
        dbg = hip.Debug()
        dbg.d_samples1 = taps["samples1"].data_ptr()
        dbg.d_feat1 = taps["feat1"].data_ptr()
        for i in range(2):
            dbg.d_pointfeat[i] = taps["pf"][i].data_ptr()
            dbg.d_knn[i] = taps["knn"][i].data_ptr()
            dbg.d_members[i] = taps["members"][i].data_ptr()
            dbg.h_gcn[i] = taps["gcn"][i].ctypes.data
        for i in range(3):
            dbg.h_dist[i] = taps["dist"][i].ctypes.data
        for i in range(4):
            dbg.h_adj[i] = taps["adj"][i].ctypes.data
    res = pipe.forward(ds, net.mode(), dbg)
    torch.cuda.synchronize()
    taps["n_adj"] = list(dbg.n_adj) if dbg is not None else None
    return res, taps, pipe


@pytest.mark.parametrize("name", ["tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"])
@pytest.mark.parametrize("mode", ["ins_infer", "sem_infer"])
def test_labels_and_metrics_match_reference_capture(golden_index, weight_sets, name, mode):
    from seggroup_amd import hip
    scene = make_fixture_scene(golden_index, name)
    g = load_golden(name)
    pre = mode[:3]
    res, _, _ = _run(scene, weight_sets[mode], mode)
    assert res.trace[:2] == [golden_index[name]["s"], golden_index[name][mode]["nclusters"][0]]
    for i in range(res.n_vectors):
        nm = hip.LABEL_NAMES[i]
        assert np.array_equal(res.labels[i], g[f"{pre}.label.{nm}"]), f"{name}/{mode}/{nm}: {int(np.sum(res.labels[i] != g[f'{pre}.label.{nm}']))} vertices differ"
    assert np.array_equal(res.iou_sem, g[f"{pre}.metric.0"])
    assert np.array_equal(res.iou_ins, g[f"{pre}.metric.1"])
    assert np.allclose(res.acc, g[f"{pre}.metric.2"], rtol=0, atol=1e-7, equal_nan=True)
    if mode == "ins_infer":
        # island_20k has an unlabeled, disconnected component holding cluster 0: the FPS-1024 fallback
        # (model.py:479-494) must have run there and nowhere else
        assert res.used_fallback == (name == "island_20k")


@pytest.mark.parametrize("name", ["tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"])
def test_stage_taps_match_oracle(golden_index, weight_sets, name):
    """Every stage tap of the pipeline vs the oracle (which tests/test_oracle_golden.py pins to the
    reference capture): adjacency lists and kNN tables bit-exact (same defined tie rule), floats
    within 1e-4 (north_star tolerance; observed ~1e-6 .. 1e-5)."""
    from oracle import cpu_ref
    scene = make_fixture_scene(golden_index, name)
    res, t, _ = _run(scene, weight_sets["ins_infer"], "ins_infer", debug=True)
    ref = cpu_ref.forward_scene(scene, weight_sets["ins_infer"], "ins_infer", keep=True)
    st = ref["stages"]
    assert res.trace == ref["trace"]
    assert np.abs(t["samples1"].cpu().numpy() - st["samples"]).max() < 1e-5
    assert np.abs(t["feat1"].cpu().numpy() - st["feat1"]).max() < FLOAT_TOL
    for i, a in enumerate((st["adj1"], st["adj2"], st["mlp_2"]["adj"], st["mlp_3"]["adj"])):
        assert t["n_adj"][i] == a.shape[0], f"adj_{i + 1} rows"
        assert np.array_equal(t["adj"][i][:a.shape[0]], a), f"adj_{i + 1}"
    for i, d in enumerate((st["d1"], st["mlp_2"]["d"], st["mlp_3"]["d"])):
        assert np.abs(t["dist"][i][:d.shape[0]] - d).max() < FLOAT_TOL, f"decision distances {i}"
    for i, nm in enumerate(("mlp_2", "mlp_3")):
        members = t["members"][i].cpu().numpy()
        knn_pos = t["knn"][i].cpu().numpy()
        knn_pts = np.empty((scene.num_points, 20), np.int64)
        knn_pts[members] = members[knn_pos]
        assert np.array_equal(knn_pts, st[nm]["knn"]), f"{nm}: kNN table (order included)"
        pf = np.empty((scene.num_points, 64), np.float32)
        pf[members] = t["pf"][i].cpu().numpy()
        assert np.abs(pf - st[nm]["point_feat"]).max() < FLOAT_TOL, nm
    C2, C3 = res.trace[1], res.trace[2]
    assert np.abs(t["gcn"][0].reshape(-1)[:C2 * 192].reshape(C2, 192) - st["mlp_2"]["gcn"]).max() < FLOAT_TOL
    assert np.abs(t["gcn"][1].reshape(-1)[:C3 * 256].reshape(C3, 256) - st["mlp_3"]["gcn"]).max() < FLOAT_TOL


def test_150k_scene_matches_reference_digests_and_oracle(golden_index, weight_sets):
    """BASELINE.json configs[1]: 150k points / 1.5k segments; labels must equal the reference capture
    (sha256 digests committed in tests/golden/index.json)."""
    from seggroup_amd import hip
    name = "scene_150k"
    scene = make_fixture_scene(golden_index, name)
    res, _, _ = _run(scene, weight_sets["ins_infer"], "ins_infer")
    e = golden_index[name]["ins_infer"]
    assert res.trace[1:5] == e["nclusters"]
    for i in range(14):
        nm = hip.LABEL_NAMES[i]
        sha = hashlib.sha256(np.ascontiguousarray(res.labels[i]).tobytes()).hexdigest()
        assert sha == e["label_sha"][nm], nm
    g = load_golden(name)
    assert np.array_equal(res.iou_sem, g["ins.metric.0"]) and np.array_equal(res.iou_ins, g["ins.metric.1"])


def test_150k_scene_float_stages_match_reference_capture_and_oracle(golden_index, weight_sets):
    """Floats at full size (VERDICT round 2, weak #1b).  scene_150k.npz holds every 64th point row of the REAL reference's MLP2 / MLP3
    outputs (capture B), its GCN outputs and decision distances, the rows of its in-cluster kNN tables that differ from the
    defined tie rule (exact score ties: tests/test_oracle_golden.py), and the oracle's GCN outputs / distances.  The HIP path must
      * produce the defined-tie-rule kNN tables bit for bit (sha256 of the full [N,20] tables) -- patched with the stored rows they
        ARE the reference's tables;
      * match the reference's point features within 1e-4 on every sampled row that is not a tie row (the split-operand EdgeConv:
        bf16 x 3 / fp16 x 2 pieces on the 16-bit matrix pipe);
      * match the oracle's GCN outputs and decision distances within 1e-4 EVERYWHERE, and the reference's everywhere EXCEPT on the
        clusters a stored tie row reaches: its own cluster, that cluster's neighbours (one GCN hop) and, in layer 3, the clusters
        that inherit a moved layer-2 cluster through the carried features."""
    name = "scene_150k"
    e = golden_index[name]
    g = load_golden(name)
    scene = make_fixture_scene(golden_index, name)
    res, t, _ = _run(scene, weight_sets["ins_infer"], "ins_infer", debug=True)
    assert res.trace[1:5] == e["ins_infer"]["nclusters"]
    stride = e["taps_stride"]
    assert np.abs(t["feat1"].cpu().numpy() - g["ins.tap.mlp_1"]).max() < FLOAT_TOL
    for i, nm in enumerate(("mlp_2", "mlp_3")):
        members = t["members"][i].cpu().numpy()
        knn_pts = np.empty((scene.num_points, 20), np.int32)
        knn_pts[members] = members[t["knn"][i].cpu().numpy()]
        assert hashlib.sha256(knn_pts.tobytes()).hexdigest() == e["knn_sha"][nm]["defined_tie_rule"], f"{nm}: kNN table"
        rows = g[f"ins.tap.knn_tie_rows.{nm}"]
        knn_pts[rows] = g[f"ins.tap.knn_tie_ref.{nm}"]
        assert hashlib.sha256(knn_pts.tobytes()).hexdigest() == e["knn_sha"][nm]["reference"], f"{nm}: kNN table vs the reference's"
        pf = np.empty((scene.num_points, 64), np.float32)
        pf[members] = t["pf"][i].cpu().numpy()
        sampled = np.arange(0, scene.num_points, stride)
        clean = ~np.isin(sampled, rows)
        d = np.abs(pf[::stride] - g[f"ins.tap.{nm}"]).max(axis=1)
        assert d[clean].max() < FLOAT_TOL, (nm, float(d[clean].max()))
        C, D = res.trace[1 + i], (192, 256)[i]
        gcn = t["gcn"][i].reshape(-1)[:C * D].reshape(C, D)
        assert np.abs(gcn - g[f"ins.oracle.gcn_{nm[-1]}"]).max() < FLOAT_TOL, nm
        # Against the REFERENCE only rows downstream of a kNN tie may deviate (round 4: the set is derived, not counted).  A tie row moves
        # its cluster's maximum; the GCN hands that to the cluster's neighbours (one hop); layer 3 also inherits, through the carried
        # features, every layer-2 cluster that moved.  cluster index = rank of the root id (model.py:759-768), rows of the GCN output.
        seg_row = 3 * (i + 1)                                             # layer_2.seg / layer_3.seg: root id per vertex (unmap is the identity here)
        roots = res.labels[seg_row]
        cl = np.searchsorted(np.unique(roots), roots)
        assert cl.max() + 1 == C
        adj = t["adj"][i + 1][:t["n_adj"][i + 1]].astype(np.int64)
        moved = np.zeros(C, bool)
        moved[cl[rows]] = True
        if i == 1:
            moved[np.unique(cl[touched_prev[cl_prev]])] = True            # layer-3 clusters that contain a layer-2 cluster which moved
        touched = moved.copy()
        touched[adj[moved[adj[:, 1]], 0]] = True
        touched[adj[moved[adj[:, 0]], 1]] = True
        bad = np.abs(gcn - g[f"ins.tap.gcn_{nm[-1]}"]).max(axis=1) >= FLOAT_TOL
        assert not np.any(bad & ~touched), (nm, int(np.sum(bad & ~touched)), "GCN rows off the reference that no kNN tie explains")
        assert bad.sum() < touched.sum()
        # the layer's decision distances: only edges with an end in a touched cluster may deviate
        want, ref = g[f"ins.oracle.dists.{i + 1}"], g[f"ins.tap.dists.{i + 1}"]
        got = t["dist"][i + 1][:want.shape[0]]
        assert np.abs(got - want).max() < FLOAT_TOL, f"decision distances {i + 1} vs the oracle"
        off = np.abs(got - ref) >= FLOAT_TOL
        assert adj.shape[0] == want.shape[0]
        assert not np.any(off & ~(touched[adj[:, 0]] | touched[adj[:, 1]])), f"decision distances {i + 1}: off the reference away from every tie"
        touched_prev, cl_prev = touched, cl
    want, ref = g["ins.oracle.dists.0"], g["ins.tap.dists.0"]                # the structural layer: no kNN-20 upstream, no exceptions
    got = t["dist"][0][:want.shape[0]]
    assert np.abs(got - want).max() < FLOAT_TOL and np.abs(got - ref).max() < FLOAT_TOL


def test_150k_scene_sem_infer_matches_reference_digests(golden_index, weight_sets):
    """sem_infer above 20k points: the 6 label vectors of the 150k fixture (weights_g1, th = 3) against the reference capture's and the
    oracle's sha256 digests."""
    from seggroup_amd import hip
    e = golden_index["scene_150k"]["sem_infer"]
    scene = make_fixture_scene(golden_index, "scene_150k")
    res, _, _ = _run(scene, weight_sets["sem_infer"], "sem_infer")
    assert e["labels_A_equal_B"] and e["oracle_equals_reference"] and res.n_vectors == 6
    assert res.trace[:2] == e["oracle_trace"] and res.trace[1] == e["nclusters"][0]
    for i in range(6):
        nm = hip.LABEL_NAMES[i]
        assert hashlib.sha256(np.ascontiguousarray(res.labels[i]).tobytes()).hexdigest() == e["label_sha"][nm], nm
    g = load_golden("scene_150k")
    assert np.array_equal(res.iou_sem, g["sem.metric.0"]) and np.array_equal(res.iou_ins, g["sem.metric.1"])


@pytest.mark.parametrize("variant", [0, 1, 2, 4])
def test_150k_scene_labels_do_not_depend_on_the_knn_kernel(golden_index, weight_sets, variant):
    """The pipeline's alternative in-cluster kNN kernels (two-pass over the chunk table; 1 / 2 / 4 waves per tile, none
    of them seeded) give the reference's labels as well (the default at this size, covered above, is one wave per tile
    with layer 3 seeded from layer 2's table)."""
    from seggroup_amd import hip
    name = "scene_150k"
    scene = make_fixture_scene(golden_index, name)
    res, _, _ = _run(scene, weight_sets["ins_infer"], "ins_infer", knn_variant=variant)
    e = golden_index[name]["ins_infer"]
    assert res.trace[1:5] == e["nclusters"]
    for i in range(14):
        nm = hip.LABEL_NAMES[i]
        assert hashlib.sha256(np.ascontiguousarray(res.labels[i]).tobytes()).hexdigest() == e["label_sha"][nm], nm


def test_stress_500k_matches_reference_and_oracle_digests(golden_index, weight_sets):
    """BASELINE.json configs[4]: 500k points / 5k segments / 20-NN.  Labels must equal the digests of the
    reference capture AND of the oracle (both committed in tests/golden/index.json; the seed was screened to be
    reference-stable: captures A and B agree, minimum decision margin 4e-5)."""
    from seggroup_amd import hip
    name = "stress_500k"
    scene = make_fixture_scene(golden_index, name)
    res, _, pipe = _run(scene, weight_sets["ins_infer"], "ins_infer")
    e = golden_index[name]["ins_infer"]
    assert e["labels_A_equal_B"] and e["oracle_equals_reference"]
    assert res.trace[1:5] == e["nclusters"] and res.trace == e["oracle_trace"]
    for i in range(14):
        nm = hip.LABEL_NAMES[i]
        sha = hashlib.sha256(np.ascontiguousarray(res.labels[i]).tobytes()).hexdigest()
        assert sha == e["oracle_label_sha"][nm], f"{nm}: HIP != oracle"
        assert sha == e["label_sha"][nm], f"{nm}: HIP != reference capture"
    g = load_golden(name)
    assert np.array_equal(res.iou_sem, g["ins.metric.0"]) and np.array_equal(res.iou_ins, g["ins.metric.1"])
    assert pipe.device_bytes() < 1 << 30          # one in-flight 500k scene needs < 1 GiB of the 288 GB


def test_determinism_same_scene_twice(golden_index, weight_sets):
    scene = make_fixture_scene(golden_index, "small_20k")
    a, _, _ = _run(scene, weight_sets["ins_infer"], "ins_infer")
    la = a.labels.copy()
    b, _, _ = _run(scene, weight_sets["ins_infer"], "ins_infer")
    assert np.array_equal(la, b.labels) and np.array_equal(a.iou_ins, b.iou_ins)


def test_results_of_consecutive_forwards_on_one_pipeline_do_not_alias(golden_index, weight_sets):
    """Pipeline.forward hands out the pinned block the library filled (no copy): a held result must survive the pipeline's later forwards, and
    the rows exported per layer on the second stream must all have landed when forward returns (both modes: 14 and 6 vectors)."""
    from seggroup_amd import hip
    from seggroup_amd.scene import DeviceScene
    small = make_fixture_scene(golden_index, "small_20k")
    tiny = make_fixture_scene(golden_index, "tiny_4k")
    first, _, pipe = _run(small, weight_sets["ins_infer"], "ins_infer")
    want = first.labels.copy()
    assert want.shape[0] == 14 and (want[12:] >= -1).all()
    d_small, d_tiny = DeviceScene.from_synthetic(small, device="cuda:0"), DeviceScene.from_synthetic(tiny, device="cuda:0")
    held = [first]
    for k in range(6):                                    # smaller scene, other mode, the same scene again
        held.append(pipe.forward(d_tiny if k % 2 == 0 else d_small, hip.MODE_SEM_INFER if k % 3 == 0 else hip.MODE_INS_INFER))
    assert np.array_equal(first.labels, want)
    again = [r for k, r in enumerate(held[1:]) if k % 2 == 1 and k % 3 != 0]
    assert again and all(np.array_equal(r.labels, want) for r in again)
    sem = [r for k, r in enumerate(held[1:]) if k % 2 == 0 and k % 3 == 0]
    ins = [r for k, r in enumerate(held[1:]) if k % 2 == 0 and k % 3 != 0]
    assert sem and ins and sem[0].n_vectors == 6 and sem[0].labels.shape[1] == d_tiny.V
    assert all(np.array_equal(r.labels, ins[0].labels) for r in ins)


def test_dropin_forward_and_driver_write_reference_files(tmp_path, golden_index, weight_sets, monkeypatch):
    """The reference's entry points: SegModel.forward(data, weak_label, info) reading the reference's
    on-disk tree by scene name, and `infer.py --ins_infer` with a reference-layout checkpoint; the 14
    txt files (one '%d\\n' per raw vertex) and their .npy twins carry the reference's integers."""
    import torch
    from seggroup_amd import hip, infer, synthetic, weights
    from seggroup_amd.data import ScanNet
    from seggroup_amd.model import SegModel
    name = "tiny_dup_4k"                                  # V != N: exercises the unmap gather
    scene = make_fixture_scene(golden_index, name)
    g = load_golden(name)
    root = str(tmp_path)
    synthetic.write_reference_tree(root, [scene])
    os_mod = __import__("os")
    ck = os_mod.path.join(root, "checkpoints", "exp", "models")
    os_mod.makedirs(ck)
    torch.save({"epoch": 6, "state_dict": weights.to_full_state_dict(weight_sets["ins_infer"]), "optimizer": {}},
               os_mod.path.join(ck, "last.t7"))
    # (1) forward() called exactly like infer.py:150-152 does
    net = SegModel(exp_name="exp", ins_infer=True, data_root=root).to("cuda:0")
    sd = torch.load(os_mod.path.join(ck, "last.t7"), map_location="cpu")["state_dict"]
    net.load_state_dict({k[len("module."):]: v for k, v in sd.items()}, strict=False)
    net.epoch = "ins_infer"
    data, weak, info = ScanNet("manual", root=root)[0]
    with torch.no_grad():
        iou_sem, iou_ins, acc = net(data[None].cuda(), weak[None].cuda(), info[None])
    net.flush()                                            # label files are written by the async writer pool
    assert iou_sem.shape == (1, 2, 40) and iou_sem.is_cuda and acc.shape == (4,)
    assert np.array_equal(iou_sem.cpu().numpy(), g["ins.metric.0"]) and np.array_equal(iou_ins.cpu().numpy(), g["ins.metric.1"])
    out_dir = os_mod.path.join(root, "results", "exp", scene.name, "ins_infer")
    for nm in hip.LABEL_NAMES:
        want = g[f"ins.label.{nm}"]
        assert [int(x) for x in open(os_mod.path.join(out_dir, nm + ".txt")).read().split()] == want.tolist(), nm
        assert np.array_equal(np.load(os_mod.path.join(out_dir, nm + ".npy")), want)
    # (2) the driver, single GPU
    for f in os_mod.listdir(out_dir):
        os_mod.remove(os_mod.path.join(out_dir, f))
    infer.main(["-n", "exp", "--ins_infer", "--root", root, "--world-size", "1"])
    assert [int(x) for x in open(os_mod.path.join(out_dir, "final.ins.txt")).read().split()] == g["ins.label.final.ins"].tolist()
    log = open(os_mod.path.join(root, "checkpoints", "exp", "run_infer.log")).read()
    assert "Network parameters: 147880" in log and "Infer(0001/0001)" in log and "==> Infer" in log
    # (3) the same driver through the per-scene SegModel.forward loop (no packs, no batching)
    for f in os_mod.listdir(out_dir):
        os_mod.remove(os_mod.path.join(out_dir, f))
    infer.main(["-n", "exp", "--ins_infer", "--root", root, "--world-size", "1", "--batch", "0"])
    assert [int(x) for x in open(os_mod.path.join(out_dir, "final.ins.txt")).read().split()] == g["ins.label.final.ins"].tolist()


def test_pipeline_larger_than_scene_and_ragged_batches(golden_index, weight_sets):
    """A pipeline (or a batch slot) sized for a bigger scene: the label vectors of a smaller scene are packed at
    ITS vertex stride (sg_result.h_labels) -- regression for reading them at the capacity stride."""
    from seggroup_amd import hip
    from seggroup_amd.model import BatchRunner, Pipeline
    from seggroup_amd.scene import DeviceScene
    w = weight_sets["ins_infer"]
    names = ["small_20k", "tiny_dup_4k", "tiny_4k", "island_20k", "tiny_dup_4k"]
    scenes = [DeviceScene.from_synthetic(make_fixture_scene(golden_index, n), device="cuda:0") for n in names]
    assert len({s.V for s in scenes}) > 2
    big = Pipeline(w, max(s.N for s in scenes) + 1000, max(s.S for s in scenes) + 10, max(s.E0 for s in scenes) + 10,
                   max(s.V for s in scenes) + 999, device="cuda:0")
    for n, sc in zip(names, scenes):
        g = load_golden(n)
        res = big.forward(sc, hip.MODE_INS_INFER)
        for i in range(14):
            assert np.array_equal(res.labels[i], g[f"ins.label.{hip.LABEL_NAMES[i]}"].astype(np.int32)), (n, i)
    big.close()
    # the engine: groups of scenes of DIFFERENT sizes advance in lock-step through batched launches (grid.y = scene); one
    # of them needs the FPS-1024 fallback, which runs for that scene alone
    for inflight, per_group in ((3, 1), (4, 4), (8, 3), (2, 2)):
        runner = BatchRunner(w, scenes, inflight=inflight, per_group=per_group, device="cuda:0")
        for _ in range(2):
            out = runner.run(scenes, hip.MODE_INS_INFER)
            for n, sc, res in zip(names, scenes, out):
                g = load_golden(n)
                assert res.labels.shape == (14, sc.V)
                for i in range(14):
                    assert np.array_equal(res.labels[i], g[f"ins.label.{hip.LABEL_NAMES[i]}"].astype(np.int32)), (n, i, inflight, per_group)
                assert np.array_equal(res.iou_ins, g["ins.metric.1"]) and np.array_equal(res.iou_sem, g["ins.metric.0"])
                assert res.used_fallback == (n == "island_20k")
        runner.close()
    # sem_infer through the engine (returns after the structural layer: 6 label vectors)
    ws = weight_sets["sem_infer"]
    runner = BatchRunner(ws, scenes, inflight=4, per_group=2, device="cuda:0")
    out = runner.run(scenes, hip.MODE_SEM_INFER)
    for n, sc, res in zip(names, scenes, out):
        g = load_golden(n)
        assert res.n_vectors == 6
        for i in range(6):
            assert np.array_equal(res.labels[i], g[f"sem.label.{hip.LABEL_NAMES[i]}"].astype(np.int32)), (n, i)
        assert np.array_equal(res.iou_sem, g["sem.metric.0"]) and np.array_equal(res.iou_ins, g["sem.metric.1"])
    runner.close()


def test_engine_pipelined_tickets_odd_scenes_and_errors(weight_sets):
    """Engine vs single pipeline on scenes the fixtures do not cover -- 2-point segments, duplicates, V != N, an island, a
    segment too large for the one-launch LDS sort (library-sort path inside a batched phase) -- with two tickets in flight
    (submit, submit, wait, wait); then a scene that exceeds the engine's capacity must fail its ticket cleanly and leave the
    engine usable."""
    import torch
    from seggroup_amd import hip, synthetic
    from seggroup_amd.model import BatchRunner, Pipeline
    from seggroup_amd.scene import DeviceScene
    W = weight_sets["ins_infer"]
    host = [synthetic.make_scene(n, s, seed, **kw) for n, s, seed, kw, mode in FUZZ if mode == "ins_infer"]
    host.append(synthetic.make_scene(30000, 6, 140, min_seg=4))                # ~5,000-point segments: beyond the LDS sort cap
    scenes = [DeviceScene.from_synthetic(h, device="cuda:0") for h in host]
    assert max(int(s.h_seg_size.max()) for s in scenes) > 2048
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    solo = Pipeline(W, *caps, device="cuda:0")
    want = [_digest(solo.forward(s, hip.MODE_INS_INFER)) for s in scenes]
    solo.close()
    eng = BatchRunner(W, scenes, inflight=6, per_group=3, device="cuda:0", timing=1)
    a, b = scenes[:5], scenes[5:]
    t1 = eng.submit(a, hip.MODE_INS_INFER)
    t2 = eng.submit(b, hip.MODE_INS_INFER)
    got = [_digest(r) for r in eng.wait(t1)] + [_digest(r) for r in eng.wait(t2)]
    assert got == want
    ms = eng.mean_stage_ms()
    assert ms["l2.knn"] > 0 and ms["l3.edgeconv.stats2"] > 0
    # capacity violation: the ticket fails, nothing hangs, the next ticket is fine
    big = DeviceScene.from_synthetic(synthetic.make_scene(caps[0] + 5000, 50, 141), device="cuda:0")
    eng2 = BatchRunner(W, scenes[:2], inflight=2, per_group=2, device="cuda:0")
    c_ok = eng2.submit(scenes[:2], hip.MODE_INS_INFER)
    assert [_digest(r) for r in eng2.wait(c_ok)] == want[:2]
    with pytest.raises(ValueError):
        eng2.submit([big], hip.MODE_INS_INFER)
    assert [_digest(r) for r in eng2.run(scenes[:2], hip.MODE_INS_INFER)] == want[:2]
    eng.close(); eng2.close()


def test_compact_label_transfer_and_growing_batches_write_the_same_files(tmp_path, weight_sets):
    """(1) sg_engine_set_label_transfer: with only the [14,S] tables crossing PCIe -- the 14 vectors looked up on the host, in the writer
    pool's workers and on first access of SceneResult.labels -- labels, metrics and every label file are identical to the full copy,
    dup / V != N scenes and sem_infer included.  (2) Ragged, GROWING batches with the writer: a submit that re-allocates
    the pinned label ring while an earlier ticket is unconsumed must not free the buffer the pool still formats files from."""
    import hashlib
    from seggroup_amd import hip, synthetic
    from seggroup_amd.model import AsyncLabelWriter, BatchRunner
    from seggroup_amd.scene import DeviceScene
    W = weight_sets["ins_infer"]
    host = [synthetic.make_scene(3000 + 700 * i, 30 + 6 * i, 86000 + i, **({"dup_frac": 0.05} if i % 3 == 0 else {})) for i in range(7)]
    scenes = [DeviceScene.from_synthetic(h, device="cuda:0") for h in host]
    assert all(s.h_seg_of_vertex is not None for s in scenes)

    def tree(root):
        out = {}
        for d in sorted(os.listdir(root)):
            for f in sorted(os.listdir(os.path.join(root, d))):
                out[d + "/" + f] = hashlib.sha256(open(os.path.join(root, d, f), "rb").read()).hexdigest()
        return out

    results = {}
    for transfer in ("full", "tables"):
        for mode, tag in ((hip.MODE_INS_INFER, "ins"), (hip.MODE_SEM_INFER, "sem")):
            root = str(tmp_path / f"{transfer}_{tag}")
            os.makedirs(root)
            eng = BatchRunner(W, scenes, inflight=4, per_group=2, device="cuda:0", label_transfer=transfer)
            wr = AsyncLabelWriter(threads=4)
            # growing batches, the second queued before the first is waited for
            t1 = eng.submit(scenes[:1], mode, writer=wr, out_dirs=[os.path.join(root, "s0")])
            t2 = eng.submit(scenes[1:4], mode, writer=wr, out_dirs=[os.path.join(root, f"s{i}") for i in (1, 2, 3)])
            r1 = eng.wait(t1)
            t3 = eng.submit(scenes[4:], mode, writer=wr, out_dirs=[os.path.join(root, f"s{i}") for i in (4, 5, 6)])
            res = r1 + eng.wait(t2) + eng.wait(t3)
            dig = [_digest(r) for r in res]
            del r1, res, t1, t2, t3                                           # nothing but the engine keeps the old label buffers alive now
            wr.flush(); wr.close(); eng.close()
            results[(transfer, tag)] = (dig, tree(root))
            assert len(results[(transfer, tag)][1]) == 7 * (28 if tag == "ins" else 12)
    for tag in ("ins", "sem"):
        assert results[("full", tag)][0] == results[("tables", tag)][0], tag
        assert results[("full", tag)][1] == results[("tables", tag)][1], tag
    # the files of the ragged run equal those of one scene at a time through a fresh engine
    eng = BatchRunner(W, scenes, inflight=2, per_group=1, device="cuda:0")
    wr = AsyncLabelWriter(threads=2)
    root = str(tmp_path / "one_by_one")
    for i, s_ in enumerate(scenes):
        eng.run([s_], hip.MODE_INS_INFER, writer=wr, out_dirs=[os.path.join(root, f"s{i}")])
    wr.flush(); wr.close(); eng.close()
    assert tree(root) == results[("full", "ins")][1]


def test_packed_scene_and_fast_driver_match_reference_capture(tmp_path, golden_index, weight_sets):
    """scene packs -> prefetching loader -> sg_batch_forward -> async writer, over a ragged
    5-scene list with batch 2 (the runner regrows once); every file carries the reference's integers and the
    summary equals the per-scene loop's."""
    import os
    import torch
    from seggroup_amd import cache, hip, infer, synthetic, weights
    names = ["tiny_4k", "small_20k", "tiny_dup_4k", "island_20k"]
    scenes = [make_fixture_scene(golden_index, n) for n in names]
    root = str(tmp_path)
    synthetic.write_reference_tree(root, scenes)
    # a pack-loaded scene is the same device scene as one staged from the arrays
    from seggroup_amd.scene import DeviceScene
    a = cache.load_pack(cache.pack_scene(root, scenes[2].name), device="cuda:0")
    b = DeviceScene.from_synthetic(scenes[2], device="cuda:0")
    for k in ("d_data", "d_adj", "d_seg_of_point", "d_seg_points", "d_seg_off", "d_unmap", "d_gt"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    for k in ("h_seg_first", "h_seg_size", "h_seg_ins", "h_seg_sem"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    ck = os.path.join(root, "checkpoints", "exp", "models")
    os.makedirs(ck)
    torch.save({"state_dict": weights.to_full_state_dict(weight_sets["ins_infer"])}, os.path.join(ck, "last.t7"))
    fast = infer.run_worker(0, 1, infer.build_parser().parse_args(
        ["-n", "exp", "--ins_infer", "--root", root, "--world-size", "1", "--batch", "2", "--inflight", "2", "-j", "2"]))
    for sc, n in zip(scenes, names):
        g = load_golden(n)
        out_dir = os.path.join(root, "results", "exp", sc.name, "ins_infer")
        for nm in hip.LABEL_NAMES:
            want = g[f"ins.label.{nm}"]
            assert np.array_equal(np.load(os.path.join(out_dir, nm + ".npy")), want), (n, nm)
            assert [int(x) for x in open(os.path.join(out_dir, nm + ".txt")).read().split()] == want.tolist(), (n, nm)
    slow = infer.run_worker(0, 1, infer.build_parser().parse_args(
        ["-n", "exp", "--ins_infer", "--root", root, "--world-size", "1", "--batch", "0"]))
    for k in fast:
        if k not in ("elapsed_s", "startup_s", "first_batch"):
            assert np.array_equal(np.asarray(fast[k]), np.asarray(slow[k]), equal_nan=True), k   # classes absent from a scene set are NaN


@pytest.mark.parametrize("n,s,seed", [(20000, 8, 77), (6000, 3, 78), (30000, 900, 79)])
def test_unusual_segmentations_match_oracle(weight_sets, n, s, seed):
    """Shapes the fixtures do not cover: a handful of huge over-segments (2.5k points each: FPS on the 16-wave
    path, kNN chunks inside one segment, clusters of 1-3 segments) and very many tiny ones (33 points each: most
    clusters at or below k = 20 neighbours after few merges).  HIP must equal the oracle computed on the spot."""
    from oracle import cpu_ref
    from seggroup_amd import hip, synthetic
    scene = synthetic.make_scene(n, s, seed, min_seg=4)
    res, _, _ = _run(scene, weight_sets["ins_infer"], "ins_infer")
    ref = cpu_ref.forward_scene(scene, weight_sets["ins_infer"], "ins_infer")
    assert res.trace == ref["trace"]
    for i in range(14):
        nm = hip.LABEL_NAMES[i]
        assert np.array_equal(res.labels[i], ref["labels"][nm].astype(np.int32)), nm
    assert np.array_equal(res.iou_sem, ref["metrics"][0]) and np.array_equal(res.iou_ins, ref["metrics"][1])


FUZZ = [  # (points, segments, seed, generator kwargs, mode): shapes the fixtures do not cover, oracle computed on the spot
    (3000, 60, 123, dict(min_seg=2), "ins_infer"),
    (2500, 120, 124, dict(min_seg=1), "ins_infer"),                                   # 2-point segments
    (5000, 25, 125, dict(dup_frac=0.1, raw_vertices=6000), "ins_infer"),              # 10 % duplicates, V != N
    (4000, 16, 126, dict(knn_edges=3), "ins_infer"),                                  # sparse graph, few big segments
    (6000, 300, 127, dict(min_seg=1, knn_edges=10), "ins_infer"),                     # dense graph, tiny segments
    (3500, 70, 128, dict(dup_frac=0.3), "ins_infer"),                                 # 30 % duplicates: many exact ties
    (8000, 90, 129, dict(island_radius=0.9), "ins_infer"),                            # unlabeled island -> FPS-1024 fallback
    (3000, 60, 130, dict(min_seg=2), "sem_infer"),
    (5000, 25, 131, dict(dup_frac=0.1, raw_vertices=4000), "sem_infer"),              # V < N
    (9000, 45, 132, dict(raw_vertices=9500), "ins_infer"),
]


@pytest.mark.parametrize("n,s,seed,kw,mode", FUZZ, ids=[f"{c[0]}x{c[1]}-{c[2]}-{c[4][:3]}" for c in FUZZ])
def test_odd_scenes_match_oracle(weight_sets, n, s, seed, kw, mode):
    from oracle import cpu_ref
    from seggroup_amd import hip, synthetic
    scene = synthetic.make_scene(n, s, seed, **kw)
    res, _, _ = _run(scene, weight_sets[mode], mode)
    ref = cpu_ref.forward_scene(scene, weight_sets[mode], mode)
    nvec = 14 if mode == "ins_infer" else 6
    assert res.trace[:len(ref["trace"])] == ref["trace"]
    for i in range(nvec):
        nm = hip.LABEL_NAMES[i]
        assert np.array_equal(res.labels[i], ref["labels"][nm].astype(np.int32)), nm
    assert np.array_equal(res.iou_sem, ref["metrics"][0]) and np.array_equal(res.iou_ins, ref["metrics"][1])
    assert bool(res.stalled) == bool(ref["stalled"])


def test_stage_timing_levels(golden_index, weight_sets):
    """sg_pipeline_set_timing: 2 = an event after every stage, 1 = only around the in-cluster kNN and the EdgeConv passes
    (what bench.py records), 0 = none; the labels do not depend on it."""
    from seggroup_amd import hip
    from seggroup_amd.model import Pipeline
    from seggroup_amd.scene import DeviceScene
    sc = DeviceScene.from_synthetic(make_fixture_scene(golden_index, "small_20k"), device="cuda:0")
    pipe = Pipeline(weight_sets["ins_infer"], sc.N, sc.S, sc.E0, sc.V, device="cuda:0")
    kernel_stages = {"l2.knn", "l3.knn", "l2.edgeconv", "l3.edgeconv", "l2.edgeconv.stats1", "l2.edgeconv.final", "l3.edgeconv.stats1",
                     "l3.edgeconv.stats2", "l3.edgeconv.final"}
    labels = {}
    for level in (2, 1, 0):
        assert pipe.set_timing(level) in (0, 1, 2)
        res = pipe.forward(sc, hip.MODE_INS_INFER)
        labels[level] = res.labels.copy()
        st = pipe.stage_times()
        nonzero = {k for k, v in st.items() if v > 0}
        if level == 2:
            assert {"contract_edges", "fps64", "mlp1", "export", "evaluate", "l2.knn", "l3.edgeconv"} <= nonzero
        elif level == 1:
            assert nonzero and nonzero <= kernel_stages and {"l2.knn", "l3.knn", "l3.edgeconv.stats2"} <= nonzero
        else:
            assert not nonzero
    assert np.array_equal(labels[2], labels[1]) and np.array_equal(labels[2], labels[0])
    assert pipe.lib.sg_pipeline_set_timing(pipe.handle, 7) < 0
    pipe.close()


@pytest.mark.parametrize("variant", [8, 0, 1])
@pytest.mark.parametrize("cfg", [FUZZ[1], FUZZ[2], FUZZ[5], FUZZ[6]], ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}")
def test_odd_scenes_with_the_large_scene_knn_kernels(weight_sets, cfg, variant):
    """The kernels the pipeline picks for large layers -- one wave per tile with layer 3 seeded from layer 2's table (8),
    two-pass over the chunk table (0), unseeded one wave per tile (1) -- forced onto small scenes with 2-point segments,
    10-30 % duplicated points and an unlabeled island: labels must still equal the oracle's."""
    from oracle import cpu_ref
    from seggroup_amd import hip, synthetic
    n, s, seed, kw, mode = cfg
    scene = synthetic.make_scene(n, s, seed, **kw)
    res, _, _ = _run(scene, weight_sets[mode], mode, knn_variant=variant)
    ref = cpu_ref.forward_scene(scene, weight_sets[mode], mode)
    assert res.trace == ref["trace"]
    for i in range(14):
        nm = hip.LABEL_NAMES[i]
        assert np.array_equal(res.labels[i], ref["labels"][nm].astype(np.int32)), nm


def _digest(res):
    h = hashlib.sha256()
    for i in range(res.n_vectors):
        h.update(np.ascontiguousarray(res.labels[i]).tobytes())
    h.update(res.iou_sem.tobytes()); h.update(res.iou_ins.tobytes()); h.update(np.nan_to_num(res.acc, nan=-1.0).tobytes())
    h.update(np.asarray(res.trace, np.int32).tobytes())
    return h.hexdigest()


def test_batch_of_64_full_size_scenes_through_the_concurrent_path(golden_index, weight_sets):
    """BASELINE.json configs[2]: 64 distinct 150k-point / 1.5k-segment scenes through the CONCURRENT path (the scene engine,
    8 groups x 8 scenes advancing through batched launches -- the shape bench.py times), twice.  Every scene's 14 label vectors, metric tensors and cluster trace
    must equal (i) the same scene through a single default-stream pipeline and (ii), for the `scene_150k` fixture seed that
    rides in the batch, the digests of the reference capture in tests/golden/index.json.  A race on a shared buffer, a
    stream-ordering slip or cross-scene state would show here and nowhere in the single-pipeline tests."""
    from seggroup_amd import hip, synthetic
    from seggroup_amd.model import BatchRunner, Pipeline
    from seggroup_amd.scene import DeviceScene
    W = weight_sets["ins_infer"]
    fixture = make_fixture_scene(golden_index, "scene_150k")
    host = [fixture] + [synthetic.make_scene(150000, 1500, 61000 + i) for i in range(63)]
    scenes = [DeviceScene.from_synthetic(s, device="cuda:0") for s in host]
    del host
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    solo = Pipeline(W, *caps, device="cuda:0")
    want = [_digest(solo.forward(s, hip.MODE_INS_INFER)) for s in scenes]
    solo.close()
    assert len(set(want)) == 64                                    # the scenes really are distinct
    runner = BatchRunner(W, scenes, inflight=80, device="cuda:0", timing=1)        # the engine in bench.py's shape: 10 groups x 8 scenes in lock-step, batches of 64
    assert (runner.groups, runner.per_group) == (10, 8)
    for rep in range(2):
        order = list(range(64)) if rep == 0 else list(range(63, -1, -1))      # second pass: other scene -> slot assignment
        res = runner.run([scenes[i] for i in order], hip.MODE_INS_INFER)
        got = [_digest(r) for r in res]
        bad = [order[j] for j in range(64) if got[j] != want[order[j]]]
        assert not bad, f"pass {rep}: scenes {bad} differ between the concurrent path and a single pipeline"
        if rep == 0:
            e = golden_index["scene_150k"]["ins_infer"]
            assert res[0].trace[1:5] == e["nclusters"]
            for i in range(14):
                nm = hip.LABEL_NAMES[i]
                assert hashlib.sha256(np.ascontiguousarray(res[0].labels[i]).tobytes()).hexdigest() == e["label_sha"][nm], nm
    runner.close()


def test_scan_far_from_the_origin_matches_oracle(weight_sets):
    """BatchNorm statistics of EdgeConv are formed from sums of y and y^2 (and, for MLP3's inner BN, from edge-feature
    moments): with absolute coordinates in the features, var = E[y^2] - mean^2 would cancel for a scan that sits ~100 m from
    the origin.  The kernels evaluate the x_i half about a point of the cloud (kernels_edgeconv.hip, "Conditioning"): labels
    must equal the float64 oracle's, point features stay within the north-star tolerance."""
    import torch
    from oracle import cpu_ref
    from seggroup_amd import hip, synthetic
    scene = synthetic.make_scene(20000, 200, 9100)
    scene.data[:, 0] += np.float32(103.0)
    scene.data[:, 1] -= np.float32(87.0)
    scene.data[:, 2] += np.float32(41.0)
    res, t, _ = _run(scene, weight_sets["ins_infer"], "ins_infer", debug=True)
    ref = cpu_ref.forward_scene(scene, weight_sets["ins_infer"], "ins_infer", keep=True)
    assert res.trace == ref["trace"]
    for i in range(14):
        nm = hip.LABEL_NAMES[i]
        assert np.array_equal(res.labels[i], ref["labels"][nm].astype(np.int32)), nm
    for i, nm in enumerate(("mlp_2", "mlp_3")):
        members = t["members"][i].cpu().numpy()
        pf = np.empty((scene.num_points, 64), np.float32)
        pf[members] = t["pf"][i].cpu().numpy()
        assert np.abs(pf - ref["stages"][nm]["point_feat"]).max() < FLOAT_TOL, nm


def _scan_book():
    import json
    import os
    from conftest import GOLDEN
    p = os.path.join(GOLDEN, "seed_scan.json")
    return json.load(open(p)) if os.path.exists(p) else {}


@pytest.mark.parametrize("workload", ["uniform_150k", "scannet_150k", "scannet_60k", "uniform_500k", "sem_uniform_150k", "sem_scannet_150k"])
def test_every_scanned_seed_matches_oracle_and_the_stable_ones_match_the_reference(weight_sets, workload):
    """tests/golden/seed_scan.json (tools/seed_scan.py, build container): for EVERY seed of a workload -- not only the ones a
    fixture screen would keep -- the digests of the oracle's 14 label vectors, of the real reference's (capture B), whether
    the reference agrees with itself (capture A == B) and the decision margins.  The engine must reproduce the oracle's
    digests on every seed, and the reference's wherever the reference is stable (A == B == oracle).  `scannet_*`: surfaces,
    10k-40k-point floor / wall segments, V != N, every other seed with 15 % exact duplicates (tiled scans).  `sem_*`: the same in sem_infer mode
    (weights_g1, th = 3: six label vectors, two layers)."""
    from seggroup_amd import hip, synthetic
    from seggroup_amd.model import BatchRunner
    from seggroup_amd.scene import DeviceScene
    book = _scan_book()
    if workload not in book:
        pytest.skip("seed scan not recorded for this workload")
    e = book[workload]
    seeds = sorted(e["seeds"], key=int)
    host = [synthetic.make_scene(e["n"], e["s"], int(sd), **e["kw"]) for sd in seeds]
    for sd, h in zip(seeds, host):                                  # the generator still produces the scanned inputs
        assert hashlib.sha256(np.ascontiguousarray(h.data).tobytes()).hexdigest() == e["seeds"][sd]["input_sha"]["data"], sd
    scenes = [DeviceScene.from_synthetic(h, device="cuda:0") for h in host]
    mode = e.get("mode", "ins_infer")
    eng = BatchRunner(weight_sets[mode], scenes, inflight=8, per_group=4, device="cuda:0")
    res = eng.run(scenes, hip.MODE_SEM_INFER if mode == "sem_infer" else hip.MODE_INS_INFER)
    stable = 0
    for sd, r in zip(seeds, res):
        rec = e["seeds"][sd]
        assert r.n_vectors == (6 if mode == "sem_infer" else 14)
        got = {hip.LABEL_NAMES[i]: hashlib.sha256(np.ascontiguousarray(r.labels[i]).tobytes()).hexdigest() for i in range(r.n_vectors)}
        assert r.trace[:len(rec["oracle_trace"])] == rec["oracle_trace"], (sd, r.trace)
        assert got == rec["oracle_label_sha"], f"seed {sd}: HIP != oracle on {[k for k in got if got[k] != rec['oracle_label_sha'][k]]}"
        if rec.get("labels_A_equal_B") and rec.get("oracle_equals_B"):
            stable += 1
            assert got == rec["reference_label_sha"], f"seed {sd}: HIP != reference"
    # uniform_500k: the reference disagrees with ITSELF (capture A != capture B) on both scanned seeds -- decision margins of 3e-5 .. 1.7e-4 at
    # 500k points sit inside the fp32 noise between its two memory layouts; HIP == oracle == capture B on both (checked above)
    assert stable >= (0 if workload == "uniform_500k" else max(1, len(seeds) // 2))
    eng.close()


@pytest.mark.parametrize("seed,kw", [(61001, {}), (61002, {}), (61003, dict(seg_profile="scannet"))], ids=["uniform-61001", "uniform-61002", "scannet-61003"])
def test_fresh_full_size_seeds_match_oracle(weight_sets, seed, kw):
    """150k points / 1.5k segments on seeds NO fixture or scan has seen, the oracle computed on the spot (16 threads: oversubscribing a 256-core
    host makes its many small NumPy calls ~10x slower): all 14 label vectors, the cluster trace and the metric tensors equal.  Kernel changes
    that only hold on the recorded seeds would show here."""
    from threadpoolctl import threadpool_limits
    import torch
    from oracle import cpu_ref
    from seggroup_amd import hip, synthetic
    scene = synthetic.make_scene(150000, 1500, seed, **kw)
    res, _, _ = _run(scene, weight_sets["ins_infer"], "ins_infer")
    torch.set_num_threads(16)
    with threadpool_limits(limits=16):
        ref = cpu_ref.forward_scene(scene, weight_sets["ins_infer"], "ins_infer")
    assert res.trace == ref["trace"]
    for i in range(14):
        nm = hip.LABEL_NAMES[i]
        assert np.array_equal(res.labels[i], ref["labels"][nm].astype(np.int32)), nm
    assert np.array_equal(res.iou_sem, ref["metrics"][0]) and np.array_equal(res.iou_ins, ref["metrics"][1])


_WALK_CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from seggroup_amd import hip, synthetic, weights
from seggroup_amd.model import BatchRunner
from seggroup_amd.scene import DeviceScene
W = weights.load_npz(os.path.join(sys.argv[1], "tests", "golden", "weights_g2.npz"))
host = [synthetic.make_scene(20000 + 3100 * i, 200 + 17 * i, 87000 + i, **({"seg_profile": "scannet"} if i == 2 else {})) for i in range(5)]
scenes = [DeviceScene.from_synthetic(h, device="cuda:0") for h in host]
eng = BatchRunner(W, scenes, inflight=8, per_group=4, device="cuda:0")
out = []
for r in eng.run(scenes, hip.MODE_INS_INFER):
    h = hashlib.sha256()
    for i in range(r.n_vectors):
        h.update(np.ascontiguousarray(r.labels[i]).tobytes())
    h.update(np.asarray(r.trace, dtype=np.int32).tobytes())
    out.append(h.hexdigest())
eng.close()
print("DIGESTS " + json.dumps(out))
"""


def test_edgeconv_walk_modes_give_the_same_labels():
    """k_edgeconv_hb's two tile walks -- the default stride and SG_EC_WALK_MODE=1 (contiguous ranges, XCD-grouped, the fused epilogue's carried
    cluster maxima doing real work) -- are development alternatives of one kernel: the label vectors of a ragged batch must not depend on
    the walk.  (The knob is read once per process: each mode runs in a child.)"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    got = {}
    for mode in ("0", "1"):
        env = dict(os.environ, SG_EC_WALK_MODE=mode)
        r = subprocess.run([sys.executable, "-c", _WALK_CHILD, ROOT], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        got[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("DIGESTS ")][-1][8:])
    assert len(got["0"]) == 5 and got["0"] == got["1"]


_OUT_OF_STEP_CHILD = r"""
import ctypes, hashlib, json, os, sys
root, cache, runs = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, root)
import numpy as np
import bench
from seggroup_amd import hip, weights
from seggroup_amd.model import Engine, Pipeline
from seggroup_amd.scene import DeviceScene
jobs = [(150000, 1500, 41000 + i, "voronoi", cache) for i in range(56)] + [(150000, 1500, 41100 + i, "scannet", cache) for i in range(8)]
it, pool = bench.generate_scenes(jobs, 16)
scenes = [DeviceScene.from_synthetic(h, device="cuda:0") for h in it]
if pool is not None:
    pool.shutdown()
W = weights.load_npz(os.path.join(root, "tests", "golden", "weights_g2.npz"))
caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
solo = Pipeline(W, *caps, device="cuda:0")
want = [bench.label_digest(solo.forward(s, hip.MODE_INS_INFER)) for s in scenes]
solo.close()
out = {"scenes": len(scenes), "wrong": {}, "results": 0}
for groups, per in ((16, 1), (6, 5), (10, 8)):
    eng = Engine(W, caps, groups=groups, per_group=per, device="cuda:0", timing=0)
    bad = []
    for rep in range(runs):
        got = [bench.label_digest(r) for r in eng.run(scenes, hip.MODE_INS_INFER)]
        bad += [(rep, i) for i in range(len(scenes)) if got[i] != want[i]]
        out["results"] += len(scenes)
    out["wrong"]["%dx%d" % (groups, per)] = bad
    eng.close()
lib = hip.lib()
if hasattr(lib, "sg_debug_knn_check"):                       # the self-checking twin (make selfcheck)
    buf = (ctypes.c_ulonglong * 136)()
    lib.sg_debug_knn_check(buf)
    out["knn_check"] = {"seeded_tiles": int(buf[2]), "lists_out_of_order_after_seeding": int(buf[0]), "at_the_output_stage": int(buf[1])}
if hasattr(lib, "sg_debug_fps_check"):
    fb = (ctypes.c_ulonglong * 8)()
    lib.sg_debug_fps_check(fb)
    out["fps_check"] = {"picks_checked": int(fb[3]), "picks_that_differ": int(fb[2]), "half_wave_maxima_that_differ": int(fb[0]), "first_indices_that_differ": int(fb[1])}
print("RESULT " + json.dumps(out), flush=True)
"""


def _out_of_step_run(tmp_path, lib_path, runs):
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    if lib_path:
        env["SEGGROUP_HIP_LIB"] = lib_path
    cache = os.path.join(os.path.dirname(str(tmp_path)), "sg_out_of_step_scenes")          # shared by the two tests of a session
    r = subprocess.run([sys.executable, "-c", _OUT_OF_STEP_CHILD, ROOT, cache, str(runs)], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_a_scenes_labels_do_not_depend_on_the_run_or_the_group_shape(tmp_path):
    """Round 5: with more groups than scenes per group (16 x 1, 6 x 5 -- the tail of every driver run, and ranks with few scenes) the groups fall out
    of step, waves of DIFFERENT kernels share a SIMD, and packed fp32 instructions (v_pk_fma_f32 in the hand-scheduled MLP2 loop, v_pk_add_f32 the
    compiler's SLP pass had put into the seeded kNN) lost results: about one scene in a hundred came out with other labels than the single pipeline
    gives, another one each run (tests/test_build.py keeps such instructions out of the library).  Round 6: 64
    full-size scenes (56 uniform + 8 ScanNet-shaped: the other size classes of FPS / sort / layout beside the rest), THREE shapes -- 16 x 1, 6 x 5
    and the bench's 10 x 8 -- three runs each = 576 scene results (the fault rate was ~1 %): every one must equal the single pipeline's."""
    got = _out_of_step_run(tmp_path, None, 3)
    assert got["scenes"] == 64 and got["results"] == 576
    assert all(not v for v in got["wrong"].values()), got["wrong"]


def test_selfcheck_library_finds_nothing(tmp_path):
    """`make selfcheck` (built by __graft_entry__.build()): the seeded kNN checks every list right after seeding and at the output stage, the
    chunk-pruned FPS of segments beyond 8,192 points re-evaluates every half-wave reduction and every pick serially -- the two places where round 5
    found results that depended on the run.  The same 64 scenes x three out-of-step shapes through that library: labels equal the single
    pipeline's, millions of lists and every pick checked, none wrong."""
    from seggroup_amd import hip
    twin = os.path.join(os.path.dirname(hip.LIB_PATH), "libseggroup_hip_selfcheck.so")
    assert os.path.exists(twin), "the self-checking twin is missing: python -c 'import __graft_entry__ as g; g.build()'"
    got = _out_of_step_run(tmp_path, twin, 2)
    assert all(not v for v in got["wrong"].values()), got["wrong"]
    k, f = got["knn_check"], got["fps_check"]
    assert k["seeded_tiles"] > 100000 and k["lists_out_of_order_after_seeding"] == 0 and k["at_the_output_stage"] == 0, k
    assert f["picks_checked"] > 1000 and f["picks_that_differ"] == 0 and f["half_wave_maxima_that_differ"] == 0 and f["first_indices_that_differ"] == 0, f


@pytest.mark.parametrize("mode_name", ["ins_infer", "sem_infer"])
def test_small_scenes_engine_equals_pipeline_in_every_group_shape(mode_name):
    """Gate for kNN changes (tools/r05_repro.py with SG_REPRO_SIZE=3000,30 as a test).  Scenes of 3,000 points / 30 segments take
    the kNN kernels with several waves per tile and clusters whose slices hold fewer than 20 candidates -- the case round 5's list-form thresholds got wrong
    (a NaN bound from a padding candidate), found by a memory fault in the two-rank bench test and by nothing in the suite.  96 such scenes through the
    engine in three shapes (lock-step 4 x 16, the bench's 10 x 8, out of step 16 x 1 and 6 x 5), twice each: every scene's labels, metric tensors and cluster
    trace equal the single pipeline's."""
    import torch
    import bench
    from conftest import ROOT
    from seggroup_amd import hip, synthetic, weights
    from seggroup_amd.model import Engine, Pipeline
    from seggroup_amd.scene import DeviceScene
    sem = mode_name == "sem_infer"
    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g1.npz" if sem else "weights_g2.npz"))
    mode = hip.MODE_SEM_INFER if sem else hip.MODE_INS_INFER
    n = 96
    # every other scene with ~20-point segments: its clusters are unions of a few small segments, the case in which a wave's slice of a
    # cluster ends up with 17-19 real candidates and a padding entry as its 20th
    scenes = [DeviceScene.from_synthetic(synthetic.make_scene(3000, 30, 40000 + i) if i % 2 == 0 else synthetic.make_scene(3000, 150, 41000 + i, min_seg=1),
                                         device="cuda:0") for i in range(n)]
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    solo = Pipeline(W, *caps, device="cuda:0")
    want = [bench.label_digest(solo.forward(s, mode)) for s in scenes]
    assert want == [bench.label_digest(solo.forward(s, mode)) for s in scenes]
    solo.close()
    for groups, per in ((4, 16), (10, 8), (16, 1), (6, 5)):
        eng = Engine(W, caps, groups=groups, per_group=per, device="cuda:0", timing=0)
        for rep in range(2):
            got = [bench.label_digest(r) for r in eng.run(scenes, mode)]
            bad = [i for i in range(n) if got[i] != want[i]]
            assert not bad, f"engine {groups} x {per}, run {rep}: scenes whose results differ from the single pipeline's: {bad}"
        eng.close()
    torch.cuda.synchronize()


_LABEL_COPY_CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import bench
from seggroup_amd import hip, synthetic, weights
from seggroup_amd.model import Engine, Pipeline
from seggroup_amd.scene import DeviceScene
W = weights.load_npz(os.path.join(sys.argv[1], "tests", "golden", "weights_g2.npz"))
host = [synthetic.make_scene(20000 + 2500 * i, 200 + 20 * i, 88000 + i, **({"dup_frac": 0.05, "raw_vertices": 26000 + 2500 * i} if i % 3 == 0 else {})) for i in range(12)]
scenes = [DeviceScene.from_synthetic(h, device="cuda:0") for h in host]
caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
eng = Engine(W, caps, groups=3, per_group=4, device="cuda:0", timing=0)
out = []
for rep in range(3):
    out.append([bench.label_digest(r) for r in eng.run(scenes, hip.MODE_INS_INFER)])
eng.close()
solo = Pipeline(W, *caps, device="cuda:0")
want = [bench.label_digest(solo.forward(s, hip.MODE_INS_INFER)) for s in scenes]
print("RESULT " + json.dumps({"runs": out, "pipeline": want}))
"""


def test_label_vectors_over_the_copy_engines_equal_the_stream_copy():
    """Round 6: the engine's label vectors leave the device through the HSA runtime's copy interface (csrc/sdma.cpp: the copy engines, issued behind the
    stream's sync) instead of hipMemcpyAsync, whose blit kernel cost the engine 6 % of its throughput.  Both ways -- the default and
    SG_ENGINE_LABEL_COPY=hip, which is also the fallback -- must hand the host the same bytes as the single pipeline does: 12 ragged scenes (V != N among
    them), three runs each (the label buffers are a ring: a copy that had not landed when the host read it would show as a stale digest)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    got = {}
    for mode in ("sdma", "hip"):
        env = dict(os.environ)
        if mode == "hip":
            env["SG_ENGINE_LABEL_COPY"] = "hip"
        else:
            env.pop("SG_ENGINE_LABEL_COPY", None)
        r = subprocess.run([sys.executable, "-c", _LABEL_COPY_CHILD, ROOT], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        got[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    for mode in got:
        for run in got[mode]["runs"]:
            assert run == got[mode]["pipeline"], mode
    assert got["sdma"]["pipeline"] == got["hip"]["pipeline"]
