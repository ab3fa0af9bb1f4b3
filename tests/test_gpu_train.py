"""SURVEY.md 8f-4 on the GPU: the train-mode tail of SegModel.forward against the capture of the real reference
(tests/golden/train_tail.npz: tools/capture_train.py, dropout pinned), and every backward entry point against torch autograd
of a float64 restatement of the same operator."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_fixture_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _classifier_from_golden(net, g):
    import torch
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.classifier.")}
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not [k for k in sd if k in missing.missing_keys]
    return {k[len("classifier."):]: v.double() for k, v in sd.items()}


@pytest.mark.parametrize("name", ["tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"])
def test_train_mode_forward_matches_reference_capture(golden_index, weight_sets, name):
    """`SegModel.forward` with neither infer flag set returns `(loss[1,2], IoU_sem, IoU_ins, acc)` like model.py:932; the loss,
    the logits and the per-instance features equal the reference's (same pinned dropout mask) within the float tolerance, the
    pseudo labels it exports on the way are the ins_infer ones."""
    import torch
    from seggroup_amd import hip
    from seggroup_amd.model import SegModel
    from seggroup_amd.scene import DeviceScene
    g = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    scene = make_fixture_scene(golden_index, name)
    net = SegModel(exp_name="t")                       # train mode: neither infer flag
    net.load_weights(weight_sets["ins_infer"])
    _classifier_from_golden(net, g)
    net.epoch = "0"
    net.dropout_keep = "pinned"
    sc = DeviceScene.from_synthetic(scene, device="cuda:0")
    res = net.pipeline_for(sc).forward(sc, hip.MODE_INS_INFER, want_feat5=True)
    assert res.feat5 is not None and res.feat5.shape[1] == 256 and res.feat5.shape[0] == res.trace[4]
    tail = net.train_tail(sc, res)
    loss = tail.forward().cpu().numpy()
    want = g[f"{name}.loss"]
    assert loss.shape == (1, 2) and loss[0, 1] == want[0, 1] == tail.K
    assert abs(loss[0, 0] - want[0, 0]) <= TOL * max(1.0, abs(want[0, 0])), (loss, want)
    assert np.abs(tail.logits.cpu().numpy() - g[f"{name}.logits"]).max() < TOL
    # Feat_6 from the tap == the reference's classifier input
    ins_gt = np.unique(res.ins5)
    feat6 = np.stack([res.feat5[res.ins5 == i].max(0) for i in ins_gt])
    assert np.abs(feat6 - g[f"{name}.feat6"]).max() < TOL
    # the labels written on the way are the inference ones
    gl = np.load(os.path.join(GOLDEN, name + ".npz"))
    for i in range(14):
        assert np.array_equal(res.labels[i], gl[f"ins.label.{hip.LABEL_NAMES[i]}"])


def test_train_tail_backward_matches_autograd(golden_index, weight_sets):
    import torch
    from oracle import cpu_ref
    from seggroup_amd.functional import TrainTail
    g = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    rng = np.random.default_rng(5)
    C, K = 37, 9
    feat5 = rng.normal(size=(C, 256)).astype(np.float32)
    ins5 = rng.integers(-1, K - 1, C); ins5[:K] = np.arange(-1, K - 1)             # every slot occupied, -1 included
    sem5 = rng.integers(0, 40, C)
    cls = {k[len("w.classifier."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.classifier.")}
    cls["bn1.weight"] = cls["bn1.weight"] + torch.from_numpy(rng.normal(size=128).astype(np.float32)) * 0.3
    cls["bn1.bias"] = torch.from_numpy(rng.normal(size=128).astype(np.float32)) * 0.2
    keep = cpu_ref.dropout_keep(K)
    tail = TrainTail(torch.from_numpy(feat5).cuda(), ins5, sem5, cls, keep)
    loss = tail.forward().cpu().numpy()
    got = {k: v.cpu().numpy() for k, v in tail.backward().items()}
    # float64 torch restatement of model.py:900-932 + util.py:12-29
    f5 = torch.from_numpy(feat5).double().requires_grad_(True)
    p = {k: v.double().clone().requires_grad_(True) for k, v in cls.items()}
    ins_gt = np.unique(ins5)
    f6 = torch.cat([torch.max(f5[np.nonzero(ins5 == i)[0]], dim=0, keepdim=True)[0] for i in ins_gt])
    gold = torch.tensor([int(sem5[np.nonzero(ins5 == i)[0][0]]) for i in ins_gt])
    h = f6 @ p["linear1.weight"].T
    y = torch.nn.functional.batch_norm(h, None, None, p["bn1.weight"], p["bn1.bias"], True, 0.1, 1e-5)
    z = torch.nn.functional.leaky_relu(y, 0.2) * torch.from_numpy(keep).double()
    logits = z @ p["linear2.weight"].T + p["linear2.bias"]
    one_hot = torch.zeros_like(logits).scatter(1, gold.view(-1, 1), 1)
    one_hot = one_hot * 0.8 + (1 - one_hot) * 0.2 / 39
    loss_sum = -(one_hot * torch.log_softmax(logits, dim=1)).sum()
    assert abs(loss[0, 0] - loss_sum.item()) < TOL * abs(loss_sum.item()) and loss[0, 1] == K
    (loss_sum / K).backward()
    for k in cls:
        w = p[k].grad.numpy()
        assert np.abs(got[k] - w).max() < TOL * max(1.0, np.abs(w).max()), k
    assert np.abs(got["feat5"] - f5.grad.numpy()).max() < TOL


def test_group_max_and_segment_max_backward():
    import torch
    from seggroup_amd.functional import aggregate_cluster_feature_backward, segment_max_backward
    rng = np.random.default_rng(6)
    R, D, G = 61, 192, 17
    rows = rng.normal(size=(R, D)).astype(np.float32)
    owner = rng.integers(0, G, R); owner[:G] = np.arange(G)
    rows[owner == 3] = rows[np.nonzero(owner == 3)[0][0]]                       # a group of identical rows: the FIRST one wins
    groups = {g_: np.nonzero(owner == g_)[0].tolist() for g_ in range(G)}
    gout = rng.normal(size=(G, D)).astype(np.float32)
    got = aggregate_cluster_feature_backward(torch.from_numpy(rows).cuda(), groups, torch.from_numpy(gout).cuda()).cpu().numpy()
    x = torch.from_numpy(rows).double().requires_grad_(True)
    out = torch.cat([torch.max(x[groups[g_]], dim=0, keepdim=True)[0] for g_ in range(G)])
    out.backward(torch.from_numpy(gout).double())
    assert np.array_equal(got, x.grad.numpy().astype(np.float32))
    # point -> cluster max over contiguous ranges
    off = np.array([0, 5, 6, 40, 41, 61], np.int32)
    gout2 = rng.normal(size=(5, D)).astype(np.float32)
    got2 = segment_max_backward(torch.from_numpy(rows).cuda(), off, torch.from_numpy(gout2).cuda()).cpu().numpy()
    x = torch.from_numpy(rows).double().requires_grad_(True)
    out = torch.cat([torch.max(x[off[c]:off[c + 1]], dim=0, keepdim=True)[0] for c in range(5)])
    out.backward(torch.from_numpy(gout2).double())
    assert np.array_equal(got2, x.grad.numpy().astype(np.float32))


@pytest.mark.parametrize("S,D,E", [(40, 192, 110), (25, 256, 0), (7, 192, 21)])
def test_gcn_backward_matches_autograd(S, D, E):
    """The GCN layer's gradients w.r.t. its input features AND through the similarity weights exp(-alpha * dist) and their row
    normalisation (model.py:262-265, 305-309, 146-151), against autograd of the reference's dense formulation."""
    import torch
    from seggroup_amd.functional import gcn_backward, gcn_forward
    rng = np.random.default_rng(S)
    X = rng.normal(size=(S, D)).astype(np.float32)
    W = (rng.normal(size=(D, D)) / np.sqrt(D)).astype(np.float32)
    pairs = {(min(a, b), max(a, b)) for a, b in rng.integers(0, S, (4 * E + 4, 2)) if a != b}
    adj = np.array(sorted(pairs)[:E], np.int64).reshape(-1, 2)
    gout = rng.normal(size=(S, D)).astype(np.float32)
    alpha = 1 / 8
    out = gcn_forward(torch.from_numpy(X).cuda(), torch.from_numpy(adj).cuda(), torch.from_numpy(W).cuda(), alpha).cpu().numpy()
    gx, gw = [t.cpu().numpy() for t in gcn_backward(torch.from_numpy(X).cuda(), torch.from_numpy(adj), torch.from_numpy(W).cuda(),
                                                     torch.from_numpy(gout).cuda(), alpha)]
    x = torch.from_numpy(X).double().requires_grad_(True)
    w = torch.from_numpy(W).double().requires_grad_(True)
    edge = torch.eye(S, dtype=torch.float64)
    if adj.shape[0]:
        a, b = torch.from_numpy(adj[:, 0]), torch.from_numpy(adj[:, 1])
        dist = torch.sqrt(((x[a] - x[b] + 1e-6) ** 2).sum(1))                    # F.pairwise_distance (model.py:272)
        sims = torch.exp(-dist * alpha)
        m = torch.zeros(S, S, dtype=torch.float64)
        m = m.index_put((a, b), sims).index_put((b, a), sims)
        edge = edge + m
    edge = edge / edge.sum(1, keepdim=True)
    ref = torch.relu((edge @ x) @ w.T)
    assert np.abs(out - ref.detach().numpy()).max() < TOL
    ref.backward(torch.from_numpy(gout).double())
    assert np.abs(gx - x.grad.numpy()).max() < TOL * max(1.0, np.abs(x.grad.numpy()).max())
    assert np.abs(gw - w.grad.numpy()).max() < TOL * max(1.0, np.abs(w.grad.numpy()).max())


def _rand_edge_problem(N, K, seed, offset=0.0):
    rng = np.random.default_rng(seed)
    x9 = rng.normal(size=(N, 9)).astype(np.float32)
    x9[:, :3] += np.float32(offset)
    x9[:, 6:] = x9[:, :3] - x9[:, :3].mean(0)
    knn = rng.integers(0, N, size=(N, K)).astype(np.int64)
    knn[:, 0] = np.arange(N)
    knn[::7, 3] = knn[::7, 2]                                    # duplicated neighbours: tied rows inside a point's k rows
    W = {"w1": rng.normal(size=(64, 18)) * 0.4, "g1": rng.normal(size=64) * 0.5 + 1.0, "b1": rng.normal(size=64) * 0.1,
         "w2": rng.normal(size=(64, 64)) * 0.2, "g2": rng.normal(size=64) * 0.5 + 1.0, "b2": rng.normal(size=64) * 0.1}
    W["g1"][5] = -0.7                                            # a negative scale: the max over k picks the smallest pre-activation
    W["g2"][9] = -0.4
    gout = rng.normal(size=(N, 64)) * (rng.random(size=(N, 64)) < 0.05)     # sparse, like the point -> cluster max leaves it
    return x9, knn, {k: v.astype(np.float32) for k, v in W.items()}, gout.astype(np.float32)


@pytest.mark.parametrize("layers,N,K,offset", [(1, 3001, 20, 0.0), (2, 3001, 20, 0.0), (2, 20000, 20, 0.0), (2, 997, 20, 500.0), (1, 400, 7, 0.0), (2, 400, 16, 0.0)])
def test_edgeconv_backward_matches_autograd(layers, N, K, offset):
    """sg_edgeconv_backward (MLP2 / MLP3 with batch-statistics BatchNorm2d, model.py:83-138) against torch.autograd over the float64
    restatement oracle.train_ref.edgeconv; gradients compared relative to each tensor's largest entry."""
    import torch
    from oracle import train_ref
    from seggroup_amd import functional as F
    x9, knn, W, gout = _rand_edge_problem(N, K, 11 + layers)
    x9[:, :3] += np.float32(offset)
    P = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in W.items()}
    args = [P["w1"], P["g1"], P["b1"]] + ([P["w2"], P["g2"], P["b2"]] if layers == 2 else [])
    pf, stats = train_ref.edgeconv(torch.tensor(x9.astype(np.float64)), torch.from_numpy(knn), *args)
    (pf * torch.tensor(gout.astype(np.float64))).sum().backward()
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    got = F.edgeconv_backward(t(x9.T[None]), t(knn[None]), t(gout.T[None]), *[t(W[k]) for k in (("w1", "g1", "b1", "w2", "g2", "b2") if layers == 2 else ("w1", "g1", "b1"))])
    torch.cuda.synchronize()
    for k in args and (("w1", "g1", "b1", "w2", "g2", "b2") if layers == 2 else ("w1", "g1", "b1")):
        want = P[k].grad.numpy()
        have = got[k].cpu().numpy().astype(np.float64)
        scale = np.abs(want).max()
        assert np.abs(have - want).max() <= 2e-4 * scale, (k, np.abs(have - want).max() / scale)
    if layers == 2:
        # the same with the second BatchNorm's statistics handed in (the training step's tape): no dense forward pass, same gradients
        bn2 = torch.cat([stats[1][0].detach(), stats[1][1].detach()]).float().to(dev)
        again = F.edgeconv_backward(t(x9.T[None]), t(knn[None]), t(gout.T[None]), *[t(W[k]) for k in ("w1", "g1", "b1", "w2", "g2", "b2")], bn2_stats=bn2)
        for k in ("w1", "g1", "b1", "w2", "g2", "b2"):
            want = P[k].grad.numpy()
            assert np.abs(again[k].cpu().numpy().astype(np.float64) - want).max() <= 2e-4 * np.abs(want).max(), ("bn2 handed in", k)
    bs = got["bn_stats"].cpu().numpy()
    assert np.abs(bs[:64] - stats[0][0].detach().numpy()).max() < 1e-4 * max(1.0, abs(offset))
    assert np.abs(bs[64:128] - stats[0][1].detach().numpy()).max() < 1e-4 * np.abs(stats[0][1].detach().numpy()).max()
    if layers == 2:
        assert np.abs(bs[128:192] - stats[1][0].detach().numpy()).max() < 1e-4
        assert np.abs(bs[192:] - stats[1][1].detach().numpy()).max() < 1e-4 * np.abs(stats[1][1].detach().numpy()).max()


@pytest.mark.parametrize("S,dup", [(37, False), (600, True)])
def test_mlp1_backward_matches_autograd(S, dup):
    """sg_mlp1_backward against torch.autograd over oracle.train_ref.mlp1 (float64), the kNN-10 table taken from the oracle.
    `dup`: clusters smaller than 64 points repeat their members (model.py:405-419), so tied samples are the normal case."""
    import torch
    from oracle import cpu_ref, train_ref
    from seggroup_amd import functional as F
    rng = np.random.default_rng(5 + S)
    samples = rng.normal(size=(S, 64, 6)).astype(np.float32)
    if dup:
        samples[::3, 32:] = samples[::3, :32]
        samples[1::5, 1:] = samples[1::5, :1]                     # a one-point segment: all 64 samples coincide
    W = {"mlp_1.conv1.0.weight": (rng.normal(size=(64, 6)) * 0.5).astype(np.float32), "mlp_1.bn1.weight": (rng.normal(size=64) * 0.5 + 1).astype(np.float32),
         "mlp_1.bn1.bias": (rng.normal(size=64) * 0.1).astype(np.float32)}
    W["mlp_1.bn1.weight"][3] = -0.6
    _, idx = cpu_ref.mlp1_forward(samples, W, return_knn=True)
    P = [torch.tensor(W[k].astype(np.float64), requires_grad=True) for k in ("mlp_1.conv1.0.weight", "mlp_1.bn1.weight", "mlp_1.bn1.bias")]
    feat, mean, var = train_ref.mlp1(torch.tensor(samples.astype(np.float64)), torch.from_numpy(idx), *P)
    gfeat = rng.normal(size=(S, 128)).astype(np.float32)
    (feat * torch.tensor(gfeat.astype(np.float64))).sum().backward()
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    got = F.mlp1_backward(t(samples.transpose(0, 2, 1)), t(gfeat), *[t(W[k]) for k in ("mlp_1.conv1.0.weight", "mlp_1.bn1.weight", "mlp_1.bn1.bias")])
    torch.cuda.synchronize()
    for k, p in zip(("w", "g", "b"), P):
        want, have = p.grad.numpy(), got[k].cpu().numpy().astype(np.float64)
        assert np.abs(have - want).max() <= 2e-4 * np.abs(want).max(), (k, np.abs(have - want).max() / np.abs(want).max())
    bs = got["bn_stats"].cpu().numpy()
    assert np.abs(bs[:64] - mean.detach().numpy()).max() < 1e-4 and np.abs(bs[64:] - var.detach().numpy()).max() < 1e-4 * var.detach().numpy().max()


def _full_state(weight_sets, g):
    """reference-keyed state: the inference fixture weights + the classifier of the golden capture"""
    from seggroup_amd import weights as Wm
    st = {k: v for k, v in Wm.to_state_dict(weight_sets["ins_infer"], prefix="").items()}
    st = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in st.items()}
    for k in g.files:
        if k.startswith("w.classifier."):
            st[k[2:]] = g[k]
    return st


@pytest.mark.parametrize("name", ["tiny_4k", "tiny_dup_4k", "small_20k", "island_20k"])
def test_training_step_gradients_match_reference_capture(golden_index, weight_sets, name):
    """The whole backward chain on HIP (tail -> Feat_5 -> GCN_3 / MLP3 -> GCN_2 / MLP2 -> MLP1) against the gradients the REAL
    reference leaves on every parameter after loss.backward() (tests/golden/train_grads.npz, tools/capture_train.py; pinned dropout
    mask).  Tolerance: relative to each tensor's largest entry; the 4k fixtures end with K = 2 instances, where BatchNorm1d over two
    rows amplifies fp32 rounding (the reference's own fp32 gradients sit 3.5e-4 from the float64 chain there)."""
    import torch
    from seggroup_amd import trainer as T
    from seggroup_amd.scene import DeviceScene
    gt = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    gg = np.load(os.path.join(GOLDEN, "train_grads.npz"))
    scene = make_fixture_scene(golden_index, name)
    sc = DeviceScene.from_synthetic(scene, device="cuda:0")
    tr = T.Trainer(_full_state(weight_sets, gt), (sc.N, sc.S, sc.E0, sc.V), device="cuda:0")
    res = tr.forward(sc)
    gl = np.load(os.path.join(GOLDEN, name + ".npz"))
    from seggroup_amd import hip
    for i in range(14):
        assert np.array_equal(res.labels[i], gl[f"ins.label.{hip.LABEL_NAMES[i]}"])
    mask = tr.dropout_mask("pinned")
    loss = tr.loss(mask)
    want_loss = gt[f"{name}.loss"]
    assert loss[0, 1] == want_loss[0, 1] and abs(loss[0, 0] - want_loss[0, 0]) <= TOL * abs(want_loss[0, 0])
    flat = tr.backward(mask).cpu().numpy().astype(np.float64)
    tol = 2e-3 if name.startswith("tiny") else 2e-4
    worst = 0.0
    for pname, off, cnt in T.param_slots():
        want = gg[f"{name}.grad.{pname}"].reshape(-1).astype(np.float64)
        err = np.abs(flat[off:off + cnt] - want).max() / np.abs(want).max()
        worst = max(worst, err)
        assert err <= tol, (pname, err)
    # running BatchNorm statistics after this one forward == the reference's buffers
    tr.update_running_stats()
    sd = tr.state_dict()
    for bname, _, _ in T.BN_LAYERS:
        for kind in ("running_mean", "running_var"):
            want = gg[f"{name}.buf.{bname}.{kind}"]
            assert np.abs(sd[f"{bname}.{kind}"].numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), (bname, kind)
        assert int(sd[f"{bname}.num_batches_tracked"]) == int(gg[f"{name}.buf.{bname}.num_batches_tracked"]) == 1
    tr.close()


@pytest.mark.parametrize("use_sgd", [True, False])
def test_optimizer_kernels_match_torch_optim(use_sgd):
    """sg_optimizer_sgd / sg_optimizer_adam on the flat vectors against torch.optim with train.py:95-99's settings, five steps"""
    import ctypes as C
    import torch
    from seggroup_amd import hip
    lib = hip.lib()
    n = 147880
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.1, momentum=0.9, weight_decay=1e-4) if use_sgd else torch.optim.Adam([ref], lr=0.001, weight_decay=1e-4)
    p = p0.clone().cuda()
    a, b = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * 0.1
        ref.grad = grad.clone()
        opt.step()
        gd = grad.cuda()
        if use_sgd:
            hip.check(lib.sg_optimizer_sgd(p.data_ptr(), gd.data_ptr(), a.data_ptr(), n, C.c_float(0.1), C.c_float(0.9), C.c_float(1e-4), int(step == 1), None))
        else:
            hip.check(lib.sg_optimizer_adam(p.data_ptr(), gd.data_ptr(), a.data_ptr(), b.data_ptr(), n, C.c_float(0.001), C.c_float(1e-4), step, None))
        torch.cuda.synchronize()
        assert (p.cpu() - ref.detach()).abs().max().item() < 2e-6


def test_reference_training_loop_body_runs_unchanged(golden_index, weight_sets, tmp_path):
    """train.py:160-168 verbatim around seggroup_amd's SegModel: `loss.backward()` runs the HIP backward chain through the autograd
    node of the returned loss and leaves the reference's gradients on the module's parameters; torch.optim.SGD then moves them."""
    import torch
    from seggroup_amd import synthetic
    from seggroup_amd.model import SegModel
    name = "small_20k"
    e = golden_index[name]
    scene = synthetic.make_scene(e["n"], e["s"], e["seed"], name="scene00000_00", **e["kw"])
    root = str(tmp_path)
    synthetic.write_reference_tree(root, [scene])
    gt = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    gg = np.load(os.path.join(GOLDEN, "train_grads.npz"))
    model = SegModel(exp_name="loop", data_root=root, out_formats=("npy",)).to("cuda:0")
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in _full_state(weight_sets, gt).items()}, strict=False)
    model.dropout_keep = "pinned"
    model.train()
    model.epoch = "1"
    optimizer = torch.optim.SGD(model.parameters(), lr=0.001 * 100, momentum=0.9, weight_decay=1e-4)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    data, weak_label, info = torch.from_numpy(scene.data)[None].cuda(), torch.from_numpy(scene.weak_label)[None].cuda(), torch.tensor([[0]])
    # ---- train.py:163-168 ----
    loss_raw, IoU_sem, IoU_ins, acc = model(data, weak_label, info)
    loss_sum = torch.sum(loss_raw[:, 0])
    loss_num = torch.sum(loss_raw[:, 1])
    loss = loss_sum / loss_num
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    # --------------------------
    assert abs(float(loss.detach()) - float(gg[f"{name}.step_loss"][0])) < 1e-4 * float(loss.detach())
    for k, p in model.named_parameters():
        want = torch.from_numpy(gg[f"{name}.grad.{k}"]).reshape(p.shape)
        assert (p.grad.cpu() - want).abs().max() <= 2e-4 * want.abs().max(), k
        moved = before[k] - 0.1 * (p.grad + 1e-4 * before[k])                 # first SGD step: buf = g + wd * p
        assert (p.detach() - moved).abs().max() < 1e-6
    for bname in ("mlp_3.bn2", "mlp_3.conv2.1", "classifier.bn1"):          # conv2.1 is the Sequential's alias of bn2 (one module)
        want = gg[f"{name}.buf.{bname.replace('conv2.1', 'bn2')}.running_var"]
        assert np.abs(model.state_dict()[f"{bname}.running_var"].cpu().numpy() - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    model.flush()
    assert os.path.exists(os.path.join(root, "results", "loop", "scene00000_00", "epoch_1", "final.ins.npy"))


@pytest.mark.parametrize("workload,seed", [("uniform_150k", 20000), ("scannet_150k", 70010)])
def test_training_step_full_size_matches_oracle_chain(workload, seed):
    """BASELINE-size scenes (150k points / 1.5k segments; uniform and ScanNet-shaped): loss and the whole gradient vector against the
    float64 oracle chain (oracle/train_ref.py -- pinned to the real reference on the small fixtures), stored by
    tools/capture_train_oracle.py because its autograd graph needs ~20 GB of host memory.  Both seeds are ones where the engine's,
    the oracle's and the reference's label vectors agree (tests/golden/seed_scan.json), so all three share the discrete structure."""
    import json
    import torch
    from seggroup_amd import synthetic, trainer as T, weights as Wm
    from seggroup_amd.scene import DeviceScene
    path = os.path.join(GOLDEN, "train_grads_full.npz")
    gf = np.load(path)
    if f"{workload}.{seed}.grad" not in gf.files:
        pytest.skip("vector not captured")
    book = json.load(open(os.path.join(GOLDEN, "seed_scan.json")))[workload]
    scene = synthetic.make_scene(book["n"], book["s"], seed, name=f"scene{seed:05d}_00", **book["kw"])
    gt = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    st = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in Wm.to_state_dict(Wm.load_npz(os.path.join(GOLDEN, "weights_g2.npz")), prefix="").items()}
    st.update({k[2:]: gt[k] for k in gt.files if k.startswith("w.classifier.")})
    sc = DeviceScene.from_synthetic(scene, device="cuda:0")
    tr = T.Trainer(st, (sc.N, sc.S, sc.E0, sc.V), device="cuda:0")
    tr.forward(sc)
    mask = tr.dropout_mask("pinned")
    loss = tr.loss(mask)
    want_loss = gf[f"{workload}.{seed}.loss"]
    assert loss[0, 1] == want_loss[1] and abs(loss[0, 0] - want_loss[0]) <= 1e-4 * want_loss[0], (loss, want_loss)
    flat = tr.backward(mask).cpu().numpy().astype(np.float64)
    want = gf[f"{workload}.{seed}.grad"].astype(np.float64)
    worst = {}
    for pname, off, cnt in T.param_slots():
        worst[pname] = np.abs(flat[off:off + cnt] - want[off:off + cnt]).max() / np.abs(want[off:off + cnt]).max()
    print(json.dumps({k: float("%.2e" % v) for k, v in worst.items()}))
    assert max(worst.values()) <= 2e-5, worst      # measured: <= 2.3e-6 on both workloads
    tr.close()


def test_train_driver_end_to_end(golden_index, tmp_path):
    """`python -m seggroup_amd.train` on a three-scene tree in the reference's on-disk layout: two epochs on one GPU, pseudo-label files
    under epoch_1 / epoch_last, the reference's log lines, checkpoints the inference driver (and SegModel, strictly) reads back, and a
    --resume that picks the optimizer state up."""
    import torch
    from seggroup_amd import infer, synthetic, train
    from seggroup_amd.model import SegModel
    root = str(tmp_path)
    scenes = []
    for i, name in enumerate(("tiny_4k", "tiny_dup_4k", "small_20k")):
        e = golden_index[name]
        scenes.append(synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{i:04d}_00", **e["kw"]))
    synthetic.write_reference_tree(root, scenes)
    args = train.build_parser().parse_args(["-n", "e2e", "--root", root, "--epochs", "2", "--out-format", "npy", "--lr", "0.0002"])
    for d in ("checkpoints/e2e/models", "results/e2e"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    r = train.run_worker(0, 1, args)
    assert r["epoch"] == 2 and r["scenes"] == 3 and np.isfinite(r["loss"])
    log = open(os.path.join(root, "checkpoints", "e2e", "run.log")).read()
    assert "Epoch[1/2](0003/0003)    Loss:" in log and "==> Epoch[2/2]" in log and "Instance mIoU (18 classes)" in log
    for tag in ("epoch_1", "epoch_last"):
        for sc in scenes:
            assert os.path.exists(os.path.join(root, "results", "e2e", sc.name, tag, "final.ins.npy")), (tag, sc.name)
    ck = torch.load(os.path.join(root, "checkpoints", "e2e", "models", "last.t7"), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 2 and len(ck["optimizer"]["state"]) == 19
    net = SegModel(exp_name="x", data_root=root)
    net.load_state_dict({k[len("module."):]: v for k, v in ck["state_dict"].items()}, strict=True)
    first = torch.load(os.path.join(root, "checkpoints", "e2e", "models", "epoch_1.t7"), map_location="cpu", weights_only=False)
    moved = max(float((ck["state_dict"][k].float() - first["state_dict"][k].float()).abs().max()) for k in ck["state_dict"] if k.endswith("weight"))
    assert moved > 0.0 and all(torch.isfinite(v.float()).all() for v in ck["state_dict"].values())
    assert int(ck["state_dict"]["module.mlp_3.bn2.num_batches_tracked"]) == 6
    # resume: one more epoch from the stored state
    args3 = train.build_parser().parse_args(["-n", "e2e", "--root", root, "--epochs", "3", "--out-format", "", "-r", "--lr", "0.0002"])
    r3 = train.run_worker(0, 1, args3)
    assert r3["epoch"] == 3
    ck3 = torch.load(os.path.join(root, "checkpoints", "e2e", "models", "last.t7"), map_location="cpu", weights_only=False)
    assert ck3["epoch"] == 3 and int(ck3["state_dict"]["module.mlp_3.bn2.num_batches_tracked"]) == 9
    # the inference driver reads what the training driver wrote
    iargs = infer.build_parser().parse_args(["-n", "e2e", "--ins_infer", "--root", root, "--batch", "0", "--out-format", "npy"])
    res = infer.run_worker(0, 1, iargs)
    assert res["n"] == 3


@pytest.mark.parametrize("smoothing", [True, False])
def test_util_cross_entropy_loss_matches_reference_formula(smoothing):
    """seggroup_amd.util.cross_entropy_loss (HIP, with autograd) against the reference's formula (util.py:12-29) in torch float64"""
    import torch
    import torch.nn.functional as F
    from seggroup_amd.util import cross_entropy_loss
    g = torch.Generator().manual_seed(4)
    pred = torch.randn(37, 40, generator=g) * 3
    gold = torch.randint(0, 40, (37,), generator=g)
    ref = pred.double().clone().requires_grad_(True)
    if smoothing:
        one_hot = torch.zeros_like(ref).scatter(1, gold.view(-1, 1), 1)
        one_hot = one_hot * (1 - 0.2) + (1 - one_hot) * 0.2 / (40 - 1)
        want = -(one_hot * F.log_softmax(ref, dim=1)).sum()
    else:
        want = F.cross_entropy(ref, gold, reduction='sum')
    (want * 0.125).backward()
    mine = pred.cuda().requires_grad_(True)
    got = cross_entropy_loss(mine, gold.cuda(), smoothing=smoothing)
    (got * 0.125).backward()
    assert abs(float(got.detach()) - float(want.detach())) < 1e-5 * float(want.detach())
    assert (mine.grad.cpu().double() - ref.grad).abs().max() < 1e-6


def test_single_instance_scene_is_refused_like_the_reference(golden_index, weight_sets):
    """One weak instance in a scene -> Feat_6 has one row and BatchNorm1d in training mode raises in the reference ('Expected more
    than 1 value per channel'); the trainer reports SG_EUNSUP at the loss instead of producing NaNs, and stays usable."""
    import copy
    from seggroup_amd import hip, trainer as T
    from seggroup_amd.scene import DeviceScene
    gt = np.load(os.path.join(GOLDEN, "train_tail.npz"))
    scene = make_fixture_scene(golden_index, "tiny_4k")
    one = copy.deepcopy(scene)
    wl = one.weak_label.copy()
    first = wl[wl[:, 1] >= 0, 1].min()
    keep = wl[:, 1] == first
    wl[~keep] = -1                                               # every click but the first instance's removed
    wl[keep, 1] = 0
    one.weak_label = wl
    tr = T.Trainer(_full_state(weight_sets, gt), (scene.data.shape[0], 4096, 1 << 20, scene.unmap.shape[0] if hasattr(scene, "unmap") else scene.data.shape[0]),
                   device="cuda:0")
    tr.forward(DeviceScene.from_synthetic(one, device="cuda:0"))
    if tr.K >= 2:
        pytest.skip("the unlabeled remainder formed its own final cluster (K = %d)" % tr.K)
    with pytest.raises(hip.SgError) as e:
        tr.loss(None)
    assert "BatchNorm1d" in str(e.value)
    # the trainer is still usable
    tr.forward(DeviceScene.from_synthetic(scene, device="cuda:0"))
    assert np.isfinite(tr.loss(tr.dropout_mask("pinned"))[0, 0])
    tr.close()


def _two_rank_worker(rank, world, root, port):
    import sys
    sys.path.insert(0, ROOT_DIR)
    from seggroup_amd import train
    args = train.build_parser().parse_args(["-n", "ddp", "--root", root, "--epochs", "1", "--out-format", "", "--backend", "gloo", "--port", str(port),
                                            "--param-digests", "--lr", "0.0002"])
    train.run_worker(rank, world, args)


ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_one_gpu_and_stay_in_step(golden_index, tmp_path):
    """The multi-GPU training path rehearsed on one GPU: two processes (gloo rendezvous on 127.0.0.1, both on cuda:0), each with its
    own Trainer and its DistributedSampler share of four scenes; after every step's single all-reduce both hold the same averaged
    gradient, so their parameter vectors must stay bit-identical to the end."""
    import socket
    import torch.multiprocessing as mp
    from seggroup_amd import synthetic
    root = str(tmp_path)
    scenes = []
    for i, name in enumerate(("tiny_4k", "tiny_dup_4k", "small_20k", "island_20k")):
        e = golden_index[name]
        scenes.append(synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{i:04d}_00", **e["kw"]))
    synthetic.write_reference_tree(root, scenes)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, root, port)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    d = os.path.join(root, "checkpoints", "ddp", "models")
    a, b = open(os.path.join(d, "rank0.sha256")).read(), open(os.path.join(d, "rank1.sha256")).read()
    assert a == b and len(a.strip()) == 64
    log = open(os.path.join(root, "checkpoints", "ddp", "run.log")).read()
    assert "Epoch[1/1](0004/0004)" in log


def test_batch_trainer_equals_the_mean_of_single_scene_steps(golden_index):
    """BatchTrainer: three scenes per optimizer step, each on its own lane / stream / host thread.  The step's
    gradient must be the MEAN of the three gradients single-scene Trainers compute for the same parameters (bit for bit: the lanes run the same
    kernels, the mean is taken in lane order), the losses and label vectors per scene the same, and the parameters after the optimizer step the
    ones that mean gives -- what DistributedDataParallel does with three ranks (train.py:88)."""
    import torch
    from seggroup_amd import synthetic, train, trainer as T, weights as Wm
    from seggroup_amd.scene import DeviceScene
    st = train.initial_state(1)
    st.update({k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in Wm.to_state_dict(Wm.load_npz(os.path.join(GOLDEN, "weights_g2.npz")), prefix="").items()})
    scenes = []
    for i, name in enumerate(("small_20k", "tiny_4k", "tiny_dup_4k")):
        e = golden_index[name]
        scenes.append(DeviceScene.from_synthetic(synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{i:04d}_00", **e["kw"]), device="cuda:0"))
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    singles, losses, labels = [], [], []
    for sc in scenes:
        tr = T.Trainer(st, caps, device="cuda:0")
        res = tr.forward(sc)
        mask = tr.dropout_mask("pinned")
        losses.append(tr.loss(mask))
        singles.append(tr.backward(mask).clone())
        labels.append(res.labels.copy())
        tr.close()
    want = (singles[0] + singles[1] + singles[2]) / 3
    bt = T.BatchTrainer(st, caps, lanes=3, device="cuda:0")
    p0 = bt.params.clone()
    for rep in range(2):                                              # twice: the lanes' buffers and streams are reused
        bt.params.copy_(p0)
        done = bt.forward_backward(scenes, keep="pinned")
        torch.cuda.synchronize()
        assert torch.equal(bt.grads, want)
        for (loss, res), l1, lab in zip(done, losses, labels):
            assert np.array_equal(loss, l1) and np.array_equal(res.labels, lab)
    # a whole step: parameters move by the SGD update of that mean gradient, the log terms are sums over the scenes
    ref = T.Trainer(st, caps, device="cuda:0")
    ref.grads.copy_(want)
    ref.optimizer_step()
    bt.params.copy_(p0)
    ls, rs, summed = bt.step(scenes, keep="pinned")
    torch.cuda.synchronize()
    assert torch.equal(bt.params, ref.params)
    assert summed[-1] == 3.0 and abs(summed[0] - sum(float(l[0, 0] / l[0, 1]) for l in losses)) < 1e-4
    with pytest.raises(ValueError):
        bt.step(scenes + scenes)                                      # more scenes than lanes
    ref.close()
    bt.close()


def test_train_driver_with_several_scenes_per_step(golden_index, tmp_path):
    """`python -m seggroup_amd.train --scenes-per-step 2` on a three-scene tree: groups of two scenes (the last group has one) through the
    BatchTrainer, the reference's log lines with the running scene count, label files for every scene, a checkpoint `infer` reads back."""
    import torch
    from seggroup_amd import synthetic, train
    root = str(tmp_path)
    scenes = []
    for i, name in enumerate(("tiny_4k", "tiny_dup_4k", "small_20k")):
        e = golden_index[name]
        scenes.append(synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{i:04d}_00", **e["kw"]))
    synthetic.write_reference_tree(root, scenes)
    args = train.build_parser().parse_args(["-n", "b2", "--root", root, "--epochs", "1", "--out-format", "npy", "--lr", "0.0002", "--scenes-per-step", "2"])
    for d in ("checkpoints/b2/models", "results/b2"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    r = train.run_worker(0, 1, args)
    assert r["epoch"] == 1 and r["scenes"] == 3 and np.isfinite(r["loss"])
    log = open(os.path.join(root, "checkpoints", "b2", "run.log")).read()
    assert "Epoch[1/1](0002/0003)    Loss:" in log and "Epoch[1/1](0003/0003)    Loss:" in log and "==> Epoch[1/1]" in log
    for sc in scenes:
        assert os.path.exists(os.path.join(root, "results", "b2", sc.name, "epoch_last", "final.ins.npy")), sc.name
    ck = torch.load(os.path.join(root, "checkpoints", "b2", "models", "last.t7"), map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.dtype.is_floating_point)
