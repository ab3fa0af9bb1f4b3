"""TEST INFRASTRUCTURE ONLY -- the checker of the training step, never the product path.

The reference's training step (model.py:900-932, train.py:160-170) is `loss.backward()` over the autograd graph of
`SegModel.forward`.  This module restates the DIFFERENTIABLE chain of that forward in torch (float64 by default) on top of
the discrete structure the oracle's own forward (`cpu_ref.forward_scene(keep=True)`) recorded -- FPS samples, kNN tables,
cluster member lists, the groups of every max-aggregation, the adjacency of every layer -- and lets torch.autograd produce
the gradients of loss = loss_sum / loss_num w.r.t. every parameter.  Nothing of it is hand-derived: it is the checker for
the hand-written HIP backward kernels.

Pinned to the real reference: tests/golden/train_grads.npz holds the gradients the unmodified reference model leaves on its
parameters for the small fixtures (tools/capture_train.py); tests/test_oracle_golden.py compares.

Each block cites the reference lines it follows.
"""
import numpy as np
import torch

from . import cpu_ref

BN_EPS = 1e-5


def _bn_lrelu(y, gamma, beta):
    """BatchNorm with batch statistics over all rows (biased variance, train mode) + LeakyReLU(0.2); y [..., C]"""
    flat = y.reshape(-1, y.shape[-1])
    mean = flat.mean(0)
    var = flat.var(0, unbiased=False)
    z = (y - mean) / torch.sqrt(var + BN_EPS) * gamma + beta
    return torch.nn.functional.leaky_relu(z, 0.2), mean, var


def _group_max(rows, groups):
    """aggregate_cluster_feature (model.py:278-288): torch.max over dim 0 of each group's rows"""
    out = []
    for g in groups:
        g = torch.as_tensor(np.asarray(g, dtype=np.int64))
        out.append(rows[g].max(dim=0)[0] if g.numel() > 1 else rows[g[0]])
    return torch.stack(out, 0)


def mlp1(samples, idx, w, g, b):
    """MLP1.forward (model.py:39-80): samples [S,64,6], idx [S,64,10] (kNN inside each sample set) -> [S,128]"""
    S, P, C = samples.shape
    nb = samples[torch.arange(S)[:, None, None], idx]                          # [S,P,10,6]
    xyz = (nb[..., :3] - nb[..., :3].mean(dim=2, keepdim=True)) * 10.0
    nb = torch.cat([xyz, nb[..., 3:]], dim=-1)
    y, mean, var = _bn_lrelu(nb @ w.T, g, b)
    y = y.max(dim=2)[0]                                                        # over the 10 neighbours
    return torch.cat([y.max(dim=1)[0], y.mean(dim=1)], dim=1), mean, var


def edgeconv(x9, knn, w1, g1, b1, w2=None, g2=None, b2=None):
    """get_graph_feature2 + MLP2 / MLP3 (model.py:83-138): x9 [N,9], knn [N,20] -> [N,64]"""
    xi = x9[:, None, :].expand(-1, knn.shape[1], -1)
    e = torch.cat([x9[knn] - xi, xi], dim=2)                                   # [N,20,18]
    h, m1, v1 = _bn_lrelu(e @ w1.T, g1, b1)
    stats = [(m1, v1)]
    if w2 is not None:
        h, m2, v2 = _bn_lrelu(h @ w2.T, g2, b2)
        stats.append((m2, v2))
    return h.max(dim=1)[0], stats


def gcn(x, adj, w, alpha=1.0 / 8.0):
    """calculate_similarity + build_similarity_matrix + GCN.forward (model.py:262-274, 305-309, 141-151)"""
    S = x.shape[0]
    adj = torch.as_tensor(np.asarray(adj, dtype=np.int64).reshape(-1, 2))
    A = torch.eye(S, dtype=x.dtype)
    if adj.shape[0]:
        d = x[adj[:, 0]] - x[adj[:, 1]] + 1e-6                                 # F.pairwise_distance's eps
        sims = torch.exp(-torch.sqrt((d * d).sum(1)) * alpha)
        A = A.index_put((adj[:, 0], adj[:, 1]), sims).index_put((adj[:, 1], adj[:, 0]), sims)
    A = A / A.sum(1, keepdim=True)
    return torch.relu((A @ x) @ w.T)


def tail(feat5, ins5, sem5, P, keep=None):
    """model.py:900-932 + Classifier (154-166) + cross_entropy_loss(smoothing=True) (util.py:12-29) -> (loss_sum, K)"""
    ins5 = np.asarray(ins5, dtype=np.int64)
    ins_gt = np.unique(ins5)
    K = int(ins_gt.shape[0])
    feat6, gold = [], []
    for ins in ins_gt:
        rows = np.nonzero(ins5 == ins)[0]
        gold.append(int(np.asarray(sem5)[rows[0]]))
        r = torch.as_tensor(rows)
        feat6.append(feat5[r].max(dim=0)[0] if rows.shape[0] > 1 else feat5[r[0]])
    feat6 = torch.stack(feat6, 0)
    h = feat6 @ P["classifier.linear1.weight"].T
    z, _, _ = _bn_lrelu(h, P["classifier.bn1.weight"], P["classifier.bn1.bias"])
    if keep is not None:
        z = z * torch.as_tensor(np.asarray(keep), dtype=z.dtype)
    logits = z @ P["classifier.linear2.weight"].T + P["classifier.linear2.bias"]
    logp = torch.log_softmax(logits, dim=1)
    t = torch.full_like(logp, 0.2 / (logp.shape[1] - 1))
    t[torch.arange(K), torch.as_tensor(gold)] = 0.8
    return -(t * logp).sum(), K


PARAM_KEYS = ("mlp_1.conv1.0.weight", "mlp_1.bn1.weight", "mlp_1.bn1.bias",
              "mlp_2.conv1.0.weight", "mlp_2.bn1.weight", "mlp_2.bn1.bias", "gcn_2.fc.weight",
              "mlp_3.conv1.0.weight", "mlp_3.bn1.weight", "mlp_3.bn1.bias", "mlp_3.conv2.0.weight", "mlp_3.bn2.weight", "mlp_3.bn2.bias",
              "gcn_3.fc.weight",
              "classifier.linear1.weight", "classifier.bn1.weight", "classifier.bn1.bias", "classifier.linear2.weight", "classifier.linear2.bias")


def training_step(scene, W, keep="pinned", dtype=torch.float64, stages=None):
    """One training step's forward + backward on one scene.  W: name -> numpy array for every key of PARAM_KEYS (conv weights
    [out,in] or [out,in,1,1]).  keep: "pinned" = cpu_ref.dropout_keep(K), None = no dropout, or a [K,128] mask.
    -> dict(loss=[loss_sum, K], step_loss, grads {name: numpy}, bn {name: (mean, biased var, rows)}, forward=<cpu_ref result>)"""
    fw = stages if stages is not None else cpu_ref.forward_scene(scene, W, mode="ins_infer", keep=True)
    st = fw["stages"]
    P = {}
    for k in PARAM_KEYS:
        a = np.asarray(W[k], dtype=np.float64)
        P[k] = torch.tensor(a.reshape(a.shape[0], -1) if a.ndim > 2 else a, dtype=dtype, requires_grad=True)
    T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=dtype)
    bn = {}

    samples = np.asarray(st["samples"], dtype=np.float32)
    _, idx1 = cpu_ref.mlp1_forward(samples, W, return_knn=True)
    feat1, m, v = mlp1(T(samples), torch.as_tensor(idx1), P["mlp_1.conv1.0.weight"], P["mlp_1.bn1.weight"], P["mlp_1.bn1.bias"])
    bn["mlp_1.bn1"] = (m, v, samples.shape[0] * 64 * 10)
    feat = _group_max(feat1, st["g21"])
    for which, gk in (("mlp_2", "gcn_2.fc.weight"), ("mlp_3", "gcn_3.fc.weight")):
        L = st[which]
        two = which == "mlp_3"
        pf, stats = edgeconv(T(L["x9"]), torch.as_tensor(np.asarray(L["knn"], dtype=np.int64)), P[f"{which}.conv1.0.weight"],
                             P[f"{which}.bn1.weight"], P[f"{which}.bn1.bias"],
                             P["mlp_3.conv2.0.weight"] if two else None, P["mlp_3.bn2.weight"] if two else None,
                             P["mlp_3.bn2.bias"] if two else None)
        rows = L["x9"].shape[0] * np.asarray(L["knn"]).shape[1]
        for i, (m, v) in enumerate(stats):
            bn[f"{which}.bn{i + 1}"] = (m, v, rows)
        seg = torch.stack([pf[torch.as_tensor(np.asarray(mm, dtype=np.int64))].max(dim=0)[0] for mm in L["members"]], 0)
        fg = gcn(torch.cat([feat, seg], dim=1), L["adj_in"], P[gk])
        feat = _group_max(fg, L["groups"])
    for groups in st["groups45"]:
        feat = _group_max(feat, groups)
    ins5, sem5 = st["ins5"], st["sem5"]
    K = int(np.unique(ins5).shape[0])
    mask = cpu_ref.dropout_keep(K) if isinstance(keep, str) else keep
    loss_sum, K = tail(feat, ins5, sem5, P, mask)
    step_loss = loss_sum / K
    step_loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().reshape(np.asarray(W[k]).shape) for k, p in P.items()}
    return dict(loss=np.array([float(loss_sum.detach()), K]), step_loss=float(step_loss.detach()), grads=grads,
                bn={k: (m.detach().numpy(), v.detach().numpy(), r) for k, (m, v, r) in bn.items()}, forward=fw, feat5=feat.detach().numpy())
