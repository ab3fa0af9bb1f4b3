"""CPU oracle for the SegGroup pseudo-label hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in NumPy, of the algorithm in the reference's
`seggroup/model.py` (all `model.py:L` citations below are relative to /root/reference/seggroup/).
It exists to check the HIP path; nothing in the product (`seggroup_amd/`) may import it.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it.

Parity pinning: every function here is checked against golden vectors captured from the real
reference running in the build container (`tools/capture_reference.py` -> `tests/golden/*.npz`,
test file `tests/test_oracle_golden.py`): integer outputs bit-exact, floats within 1e-5 of the
contiguous-patched capture "B" (SURVEY.md 7.3-0).

Numerics policy:
  * integer / ordering semantics (union order, member order, FPS picks, kNN ranks, argmin ties)
    follow the reference literally;
  * the two places where fp32 rounding decides an integer result are evaluated with the
    reference's exact fp32 operation order: kNN scores (model.py:30-36, recipe in SURVEY.md 7.3-2)
    and FPS distances (model.py:319-326);
  * all other float math (conv / BatchNorm batch statistics / GCN / distances) is evaluated in
    float64 and rounded to float32 once -- an "exact" evaluation that the reference's own fp32
    results approach to ~2e-6 and that is thread-count independent.

`faithful=True` selects the reference's per-element Python loops where a vectorised NumPy
equivalent exists (edge contraction, export); both give identical results (tested) -- the faithful
mode is what `bench.py` times as the CPU baseline ("port").
"""
from __future__ import annotations

import numpy as np

SEM_VALID_CLASS_IDS = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
INS_VALID_CLASS_IDS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])

BN_EPS = 1e-5          # nn.BatchNorm2d default (model.py:68)
LRELU = 0.2            # model.py:71
PAIR_EPS = 1e-6        # F.pairwise_distance default eps (model.py:273)
F32 = np.float32


# ------------------------------------------------------------------------------------------------
# A.1 / A.2  partition state  (model.py:169-214, 712-721)
# ------------------------------------------------------------------------------------------------
class Partition:
    """Quick-find partition of the N points keyed by point index.

    `root[p]`       current cluster id of point p (= index of the first point of the over-segment
                    that currently roots the cluster)
    `chunks[r]`     ordered member list of root r, kept as a list of arrays (concatenation order is
                    load-bearing: it fixes FPS start points and the n<=k kNN rows)
    `ins/sem/npts`  per-root weak instance / semantic label and point count (stale on dead roots,
                    exactly like the reference leaves them)
    """

    def __init__(self, weak_ins, weak_sem, seg):
        seg = np.asarray(seg, dtype=np.int64)
        n = seg.shape[0]
        self.n = n
        self.ins = np.array(weak_ins, dtype=np.int64).copy()
        self.sem = np.array(weak_sem, dtype=np.int64).copy()
        self.npts = np.ones(n, dtype=np.float64)
        self.root = np.arange(n, dtype=np.int64)
        self.chunks = {}
        order = np.argsort(seg, kind="stable")
        counts = np.bincount(seg) if n else np.zeros(0, np.int64)
        pos = 0
        for c in counts:
            if c == 0:
                continue
            m = order[pos:pos + c]
            r = int(m[0])
            self.root[m] = r
            self.npts[r] = float(c)
            self.chunks[r] = [m]
            pos += c

    def members(self, r):
        ch = self.chunks[r]
        if len(ch) > 1:
            ch = [np.concatenate(ch)]
            self.chunks[r] = ch
        return ch[0]

    def union(self, a, b):
        """Merge root a INTO root b (model.py:181-192). Returns True if points moved."""
        a, b = int(a), int(b)
        if a == b:
            return False
        ia, ib = self.ins[a], self.ins[b]
        if ia != -1 and ib != -1 and ia != ib:
            return False                                   # label veto
        moved = a in self.chunks
        if moved:
            for m in self.chunks[a]:
                self.root[m] = b
        self.npts[b] += self.npts[a]
        if ia != ib:
            self.ins[b] = -ia * ib
            self.sem[b] = -self.sem[a] * self.sem[b]
        if moved:
            self.chunks.setdefault(b, []).extend(self.chunks.pop(a))
        return moved

    def roots(self):
        """Live roots in ascending order == cluster numbering of get_cluster_list (model.py:209-214)."""
        return np.array(sorted(self.chunks.keys()), dtype=np.int64)

    def clusters(self):
        rs = self.roots()
        return rs, [self.members(int(r)) for r in rs]


class Layer:
    """Frozen numbering of one grouping layer: cluster c <-> root `unmap[c]`."""

    def __init__(self, part: Partition):
        self.unmap, self.members = part.clusters()
        self.count = len(self.unmap)
        self.map = {int(r): i for i, r in enumerate(self.unmap)}

    def parents_of(self, older: "Layer", part: Partition):
        """For each cluster of this (newer) layer, the older-layer clusters it absorbed, in
        older-layer order (model.py:766-768)."""
        new_of_old = np.array([self.map[int(part.root[r])] for r in older.unmap], dtype=np.int64)
        groups = [[] for _ in range(self.count)]
        for j, c in enumerate(new_of_old):
            groups[c].append(j)
        return groups, new_of_old


# ------------------------------------------------------------------------------------------------
# A.3  edge contraction  (model.py:291-302)
# ------------------------------------------------------------------------------------------------
def contract_edges(adj_old, part: Partition, unmap_old, layer_new: Layer, faithful=False):
    adj_old = np.asarray(adj_old, dtype=np.int64).reshape(-1, 2)
    if adj_old.shape[0] == 0:
        return np.zeros((0, 2), dtype=np.int64)
    unmap_old = np.asarray(unmap_old, dtype=np.int64)
    if faithful:
        rows = []
        for a, b in adj_old.tolist():
            ia = layer_new.map[int(part.root[unmap_old[a]])]
            ib = layer_new.map[int(part.root[unmap_old[b]])]
            if ia == ib:
                continue
            rows.append((ia, ib))
        new = np.array(rows, dtype=np.int64).reshape(-1, 2)
    else:
        lut = np.full(part.n, -1, dtype=np.int64)
        lut[layer_new.unmap] = np.arange(layer_new.count)
        ia = lut[part.root[unmap_old[adj_old[:, 0]]]]
        ib = lut[part.root[unmap_old[adj_old[:, 1]]]]
        keep = ia != ib
        new = np.stack([ia[keep], ib[keep]], axis=1)
    if new.shape[0] == 0:
        return np.zeros((0, 2), dtype=np.int64)
    new = np.sort(new, axis=1)
    return np.unique(new, axis=0)


# ------------------------------------------------------------------------------------------------
# A.4  farthest point sampling + cluster sampling  (model.py:319-426)
# ------------------------------------------------------------------------------------------------
def _sqdist32(p, pts):
    """((p - pts)**2).sum(-1) in fp32 with individually rounded squares, (dx2+dy2)+dz2."""
    d = pts - p
    d = d * d
    return (d[:, 0] + d[:, 1]) + d[:, 2]


def fps(pts, k):
    """farthest_point_sampling(pts, k, initial_idx=0, skip_initial=True) (model.py:329-395)."""
    pts = np.ascontiguousarray(pts, dtype=F32)
    idx = np.zeros(k, dtype=np.int64)
    mind = _sqdist32(pts[0], pts)
    idx[0] = int(np.argmax(mind))                 # first index on ties
    mind = _sqdist32(pts[idx[0]], pts)            # min-dist array RESET to this pick (model.py:386)
    for i in range(1, k):
        idx[i] = int(np.argmax(mind))
        mind = np.minimum(mind, _sqdist32(pts[idx[i]], pts))
    return idx


def fps_general(pts, k, initial_idx=0, skip_initial=False):
    """farthest_point_sampling with any start and either skip_initial (model.py:369-394): (indices [k], distances [k,n]); l2_norm as NumPy
    evaluates ((x - y)**2).sum(axis=-1) over a short axis: squares rounded one by one, added left to right."""
    pts = np.ascontiguousarray(pts, dtype=F32)

    def dist_to(p):
        d = pts - p
        d = d * d
        out = d[:, 0].copy()
        for c in range(1, d.shape[1]):
            out = out + d[:, c]
        return out
    idx = np.zeros(k, dtype=np.int64)
    dist = np.zeros((k, pts.shape[0]), dtype=F32)
    idx[0] = initial_idx
    mind = dist_to(pts[idx[0]])
    if skip_initial:
        idx[0] = int(np.argmax(mind))
        mind = dist_to(pts[idx[0]])
    dist[0] = mind
    for i in range(1, k):
        idx[i] = int(np.argmax(mind))
        d = dist_to(pts[idx[i]])
        dist[i] = d
        mind = np.minimum(mind, d)
    return idx, dist


def fps_with_fixup(pts, k):
    """FPS + the trailing-zero fix-up of get_cluster_pointcloud (model.py:407-412)."""
    ch = fps(pts, k)
    if ch[-1] == 0:
        j = 1
        while j <= k and ch[-j] == 0:
            j += 1
        # reference loop: j runs 1..k, breaks at first non-zero from the end; no break -> j == k
        j = min(j, k)
        invalid = j - 1
        if invalid > 0:
            ch[-invalid:] = ch[:invalid]
    return ch


def sample_indices(members, xyz, P):
    """Point indices of one cluster's P samples: members tiled P//n times, then P%n FPS picks."""
    n = members.shape[0]
    rep, rem = P // n, P % n
    parts = [np.tile(members, rep)] if rep else []
    if rem > 0:
        parts.append(members[fps_with_fixup(xyz[members], rem)])
    return np.concatenate(parts)


def sample_clusters(data, layer: Layer, P, transform=True):
    """get_cluster_pointcloud (model.py:398-426) -> ([S,P,C] f32, [S,P] indices)."""
    data = np.asarray(data, dtype=F32)
    xyz = np.ascontiguousarray(data[:, :3])
    out = np.empty((layer.count, P, data.shape[1]), dtype=F32)
    sel = np.empty((layer.count, P), dtype=np.int64)
    for c, m in enumerate(layer.members):
        ii = sample_indices(m, xyz, P)
        sel[c] = ii
        blk = data[ii].copy()
        if transform:
            # torch's fp32 `cluster_data[:, :3].mean(0)` (model.py:421): four interleaved accumulators over the rows, added in
            # order, divided by P (probed against torch 2.10 CPU: bit-equal on every column).  Matters when all samples of a
            # segment coincide: the reference then normalises the rounding error of this sum, a float64 mean yields 0 / 0.
            acc = [np.zeros(3, F32) for _ in range(4)]
            for r in range(P):
                acc[r & 3] = (acc[r & 3] + blk[r, :3]).astype(F32)
            mean = ((((acc[0] + acc[1]).astype(F32) + acc[2]).astype(F32) + acc[3]).astype(F32) / F32(P)).astype(F32)
            blk[:, :3] = blk[:, :3] - mean
            with np.errstate(divide="ignore", invalid="ignore"):
                blk[:, :3] = blk[:, :3] / np.abs(blk[:, :3]).max()
        out[c] = blk
    return out, sel


# ------------------------------------------------------------------------------------------------
# A.8 / A.9  kNN  (model.py:30-36, 512-522)
# ------------------------------------------------------------------------------------------------
def knn_scores(xq, xall):
    """fp32 score rows s[i,j] = ((-xx_j) - inner_ij) - xx_i for queries xq against xall.

    inner = -2 * fma(z_i,z_j, fma(y_i,y_j, fl(x_i*x_j))) (MKL K=3 dot product), xx = fl(fl(x2+y2)+z2)
    with separately rounded squares.  The fma is emulated as a float64
    multiply-add rounded to fp32 (product exact in fp64; double rounding differs from a true fma
    with probability ~2^-29 per op, irrelevant for ranks).  Elementwise torch ops (multi-threaded);
    none of them contracts into an FMA.
    """
    import torch
    q = torch.from_numpy(np.ascontiguousarray(xq, dtype=F32))
    a = torch.from_numpy(np.ascontiguousarray(xall, dtype=F32))
    qd, ad = q.double(), a.double()
    t = q[:, 0:1] * a[:, 0][None, :]
    t = torch.addcmul(t.double(), qd[:, 1:2], ad[:, 1][None, :]).float()
    t = torch.addcmul(t.double(), qd[:, 2:3], ad[:, 2][None, :]).float()
    inner = t * -2.0

    def sq(x):
        s = x * x
        return (s[:, 0] + s[:, 1]) + s[:, 2]
    xx_q, xx_a = sq(q), sq(a)
    return (((-xx_a)[None, :] - inner) - xx_q[:, None]).numpy()


def topk_desc(scores, k):
    """Indices of the k largest entries per row, descending.

    torch.topk leaves the order of EQUAL scores unspecified (on CPU it falls out of libstdc++'s
    partial_sort / nth_element internals).  Exact fp32 score ties between different points do occur
    (the expanded form -|a|^2+2ab-|b|^2 has ~1e-5 absolute resolution), so the build DEFINES the
    rule: among equal scores the lower index wins, output ordered by (score desc, index asc).  The
    HIP kernels implement the same rule, which makes HIP-vs-oracle comparisons exact; against the
    reference capture the rule can only differ on rows with a tie at rank k (tests/test_oracle_golden.py
    checks exactly that).
    """
    import torch
    s = np.ascontiguousarray(scores, dtype=F32)
    st = torch.from_numpy(s)
    vals_t, idx_t = st.topk(k, dim=-1)                                   # the k largest VALUES are well defined
    vals, idx = vals_t.numpy(), idx_t.numpy()
    vk = vals[:, -1]
    tie = (st >= vals_t[:, -1:]).sum(dim=1).numpy() > k                  # a tie straddles rank k
    for r in np.nonzero(tie)[0]:                                         # rare: apply the index rule explicitly
        row = s[r]
        gt = np.nonzero(row > vk[r])[0]
        eq = np.nonzero(row == vk[r])[0][:k - gt.size]
        sel = np.concatenate([gt, eq])
        idx[r] = sel
        vals[r] = row[sel]
    order = np.lexsort((idx, -vals), axis=1)                             # score desc, index asc among ties
    return np.take_along_axis(idx, order, axis=1)


def knn_local(x, k, chunk=2048):
    """knn() on one point set x[n,C<=3] -> [n,k] local indices."""
    n = x.shape[0]
    out = np.empty((n, k), dtype=np.int64)
    for s in range(0, n, chunk):
        out[s:s + chunk] = topk_desc(knn_scores(x[s:s + chunk], x), k)
    return out


def cluster_knn(xyz, layer: Layer, k=20):
    """get_knn (model.py:512-522): in-cluster kNN; clusters with n <= k list all members in member
    order and leave the remaining columns 0 (= global point 0)."""
    xyz = np.asarray(xyz, dtype=F32)
    table = np.zeros((xyz.shape[0], k), dtype=np.int64)
    for m in layer.members:
        n = m.shape[0]
        if k >= n:
            table[m, :n] = m[None, :]
        else:
            table[m] = m[knn_local(xyz[m], k)]
    return table


# ------------------------------------------------------------------------------------------------
# A.6  MLP1  (model.py:39-80)
# ------------------------------------------------------------------------------------------------
def _bn_lrelu(y, gamma, beta, mean, var):
    """BatchNorm (given batch statistics) + LeakyReLU, in place on the float64 block `y`."""
    scale = gamma / np.sqrt(var + BN_EPS)
    y *= scale
    y += beta - mean * scale
    np.maximum(y, LRELU * y, out=y)      # slope < 1: LeakyReLU(y) = max(y, 0.2 y)
    return y


def mlp1_forward(samples, W, return_knn=False):
    """samples [S,64,6] f32 -> Feat_1 [S,128] f32."""
    S, P, C = samples.shape
    x = samples.astype(F32)
    idx = np.empty((S, P, 10), dtype=np.int64)
    for s in range(S):
        idx[s] = topk_desc(knn_scores(x[s, :, :3], x[s, :, :3]), 10)
    nb = np.take_along_axis(x[:, None, :, :].astype(np.float64), idx[..., None].reshape(S, P * 10, 1).repeat(C, 2)[:, None],
                            axis=2).reshape(S, P, 10, C)
    nb[..., :3] = (nb[..., :3] - nb[..., :3].mean(axis=2, keepdims=True)) * 10.0
    w = W["mlp_1.conv1.0.weight"].astype(np.float64)
    y = nb.reshape(-1, C) @ w.T
    mean, var = y.mean(0), y.var(0)
    y = _bn_lrelu(y, W["mlp_1.bn1.weight"].astype(np.float64), W["mlp_1.bn1.bias"].astype(np.float64), mean, var)
    y = y.reshape(S, P, 10, 64).max(axis=2)
    feat = np.concatenate([y.max(axis=1), y.mean(axis=1)], axis=1).astype(F32)
    return (feat, idx) if return_knn else feat


# ------------------------------------------------------------------------------------------------
# A.7  EdgeConv MLP2 / MLP3  (model.py:83-138) and per-cluster centring (model.py:429-436)
# ------------------------------------------------------------------------------------------------
def centre_per_cluster(data, layer: Layer):
    """combine_centralized_pointcloud: [N,6] -> [N,9] with XYZ - mean XYZ of the point's cluster."""
    data = np.asarray(data, dtype=F32)
    cen = data[:, :3].copy()
    for m in layer.members:
        mu = data[m, :3].astype(np.float64).mean(0).astype(F32)
        cen[m] = data[m, :3] - mu
    return np.concatenate([data, cen], axis=1)


def _edge_rows(x9, idx, lo, hi):
    xi = x9[lo:hi].astype(np.float64)
    xj = x9[idx[lo:hi]].astype(np.float64)
    k = idx.shape[1]
    xi = np.broadcast_to(xi[:, None, :], xj.shape)
    return np.concatenate([xj - xi, xi], axis=2).reshape(-1, 18), k


def edgeconv_forward(x9, idx, W, which, chunk=2048):
    """MLP2 (`which`='mlp_2', one conv) or MLP3 ('mlp_3', two convs) -> [N,64] f32.
    BatchNorm uses batch statistics over all N*k rows (train mode, model.py:109,124,128)."""
    N = x9.shape[0]
    w1 = W[f"{which}.conv1.0.weight"].astype(np.float64)
    g1, b1 = W[f"{which}.bn1.weight"].astype(np.float64), W[f"{which}.bn1.bias"].astype(np.float64)
    two = which == "mlp_3"
    if two:
        w2 = W["mlp_3.conv2.0.weight"].astype(np.float64)
        g2, b2 = W["mlp_3.bn2.weight"].astype(np.float64), W["mlp_3.bn2.bias"].astype(np.float64)
    s1 = np.zeros(64)
    q1 = np.zeros(64)
    rows = 0
    for lo in range(0, N, chunk):
        e, _ = _edge_rows(x9, idx, lo, min(N, lo + chunk))
        y = e @ w1.T
        s1 += y.sum(0)
        q1 += (y * y).sum(0)
        rows += y.shape[0]
    m1 = s1 / rows
    v1 = q1 / rows - m1 * m1
    if two:
        s2 = np.zeros(64)
        q2 = np.zeros(64)
        for lo in range(0, N, chunk):
            e, _ = _edge_rows(x9, idx, lo, min(N, lo + chunk))
            z = _bn_lrelu(e @ w1.T, g1, b1, m1, v1) @ w2.T
            s2 += z.sum(0)
            q2 += (z * z).sum(0)
        m2 = s2 / rows
        v2 = q2 / rows - m2 * m2
    out = np.empty((N, 64), dtype=F32)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        e, k = _edge_rows(x9, idx, lo, hi)
        h = _bn_lrelu(e @ w1.T, g1, b1, m1, v1)
        if two:
            h = _bn_lrelu(h @ w2.T, g2, b2, m2, v2)
        out[lo:hi] = h.reshape(hi - lo, k, 64).max(axis=1).astype(F32)
    return out


def group_max(rows, groups):
    """aggregate_cluster_feature (model.py:278-288): element-wise max over each group's rows."""
    return np.stack([rows[np.asarray(g, dtype=np.int64)].max(axis=0) for g in groups], axis=0)


# ------------------------------------------------------------------------------------------------
# A.10  edge distance / similarity / GCN  (model.py:262-274, 305-309, 141-151)
# ------------------------------------------------------------------------------------------------
def edge_distance(feat, adj):
    adj = np.asarray(adj, dtype=np.int64).reshape(-1, 2)
    f = feat.astype(np.float64)
    d = f[adj[:, 0]] - f[adj[:, 1]] + PAIR_EPS
    return np.sqrt((d * d).sum(axis=1)).astype(F32)


def gcn_forward(feat, adj, Wfc, alpha=1.0 / 8.0):
    adj = np.asarray(adj, dtype=np.int64).reshape(-1, 2)
    S = feat.shape[0]
    sims = np.exp(-edge_distance(feat, adj).astype(np.float64) * alpha)
    A = np.eye(S)
    A[adj[:, 0], adj[:, 1]] = sims
    A[adj[:, 1], adj[:, 0]] = sims
    A = A / A.sum(axis=1, keepdims=True)
    out = (A @ feat.astype(np.float64)) @ Wfc.astype(np.float64).T
    return np.maximum(out, 0.0).astype(F32)


# ------------------------------------------------------------------------------------------------
# A.5  group_nearby_clusters  (model.py:218-258)
# ------------------------------------------------------------------------------------------------
def group_nearby(part: Partition, dist, adj, layer: Layer, th):
    """Returns (connected_mask[E], stalled).  `stalled` is True where the reference would loop
    forever (a <5-point cluster whose every merge is vetoed); here the sweep loop
    stops after a sweep that saw a small endpoint but moved no point."""
    adj = np.asarray(adj, dtype=np.int64).reshape(-1, 2)
    ra = layer.unmap[adj[:, 0]] if adj.size else np.zeros(0, np.int64)
    rb = layer.unmap[adj[:, 1]] if adj.size else np.zeros(0, np.int64)
    E = adj.shape[0]
    dist = np.asarray(dist, dtype=F32)
    thf = F32(th)
    for e in range(E):
        if dist[e] > thf:
            continue
        part.union(part.root[ra[e]], part.root[rb[e]])
    stalled = False
    while True:
        small = False
        moved = False
        for e in range(E):
            a, b = part.root[ra[e]], part.root[rb[e]]
            if part.npts[a] < 5 or part.npts[b] < 5:
                moved |= part.union(a, b)
                small = True
        if not small:
            break
        if not moved:
            stalled = True
            break
    connected = part.root[ra] == part.root[rb]
    return connected, stalled


# ------------------------------------------------------------------------------------------------
# A.12  group_unlabeled_clusters  (model.py:439-509)
# ------------------------------------------------------------------------------------------------
def group_unlabeled(part: Partition, feat, adj, layer: Layer, data, faithful=False, record=None):
    """`record` (a list): receives the row groups of every max-aggregation made here, in order (the training tape)."""
    old = layer
    adj = np.asarray(adj, dtype=np.int64).reshape(-1, 2)
    count_old = feat.shape[0]
    while True:
        S = feat.shape[0]
        D = np.full((S, S), F32(1000.0), dtype=F32)
        if adj.shape[0]:
            d = edge_distance(feat, adj)
            D[adj[:, 0], adj[:, 1]] = d
            D[adj[:, 1], adj[:, 0]] = d
        nearest = np.argmin(D, axis=1)            # first index on ties (torch.min on CPU)
        for i in range(S):
            r1 = part.root[old.unmap[i]]
            if part.ins[r1] != -1:
                continue
            part.union(r1, part.root[old.unmap[nearest[i]]])
        new = Layer(part)
        groups, _ = new.parents_of(old, part)
        adj = contract_edges(adj, part, old.unmap, new, faithful)
        feat = group_max(feat, groups)
        if record is not None:
            record.append(groups)
        old = new
        if feat.shape[0] == count_old:
            break
        count_old = feat.shape[0]

    any_unlabeled = any(part.ins[part.root[r]] == -1 for r in old.unmap)
    if any_unlabeled:
        samples, _ = sample_clusters(np.asarray(data, dtype=F32)[:, :3], old, 1024, transform=False)
        for i in range(old.count):
            r1 = part.root[old.unmap[i]]
            if part.ins[part.root[r1]] != -1:
                continue
            m = samples[i].astype(np.float64).mean(0).astype(F32)
            dd = samples - m
            dd = dd * dd
            dmin = ((dd[..., 0] + dd[..., 1]) + dd[..., 2]).min(axis=1)
            for j in np.argsort(dmin, kind="stable").tolist():
                if j == i:
                    continue
                r2 = part.root[old.unmap[j]]
                if part.ins[part.root[r2]] == -1:
                    continue
                part.union(r1, r2)
        new = Layer(part)
        groups, _ = new.parents_of(old, part)
        adj = contract_edges(adj, part, old.unmap, new, faithful)
        feat = group_max(feat, groups)
        if record is not None:
            record.append(groups)
    return feat, adj


# ------------------------------------------------------------------------------------------------
# 8f-4  train-mode tail (forward; the gradients are oracle/train_ref.py): per-instance features -> Classifier -> label-smoothed CE  (model.py:900-932,
# 154-166; util.py:12-29).  Dropout is PINNED: `keep` [K,128] is the mask already scaled by 1 / (1 - p).
# ------------------------------------------------------------------------------------------------
def dropout_keep(K, seed=97):
    """The pinned dropout mask both sides use: element kept (x2) iff its counter-based uniform is < 0.5."""
    from seggroup_amd.synthetic import uniform01
    return np.where(uniform01(seed, int(K), int(K) * 128).reshape(int(K), 128) < 0.5, 2.0, 0.0).astype(F32)


def train_tail(feat5, ins5, sem5, Wc, keep=None):
    """-> dict(loss=[loss_sum, K], logits [K,40], feat6 [K,256], group [C], gold [K]); float64 arithmetic."""
    ins5 = np.asarray(ins5, dtype=np.int64)
    ins_gt = np.unique(ins5)                                      # model.py:909 (sorted; -1 is a label like any other)
    group = np.searchsorted(ins_gt, ins5).astype(np.int32)
    K = ins_gt.shape[0]
    feat6 = np.stack([np.asarray(feat5, dtype=np.float64)[group == k].max(axis=0) for k in range(K)])
    gold = np.array([np.asarray(sem5)[np.nonzero(group == k)[0][0]] for k in range(K)], dtype=np.int64)
    h = feat6 @ Wc["classifier.linear1.weight"].astype(np.float64).T
    mean, var = h.mean(0), h.var(0)
    y = (h - mean) / np.sqrt(var + BN_EPS) * Wc["classifier.bn1.weight"].astype(np.float64) + Wc["classifier.bn1.bias"].astype(np.float64)
    z = np.maximum(y, LRELU * y)
    if keep is not None:
        z = z * np.asarray(keep, dtype=np.float64)
    logits = z @ Wc["classifier.linear2.weight"].astype(np.float64).T + Wc["classifier.linear2.bias"].astype(np.float64)
    lse = np.log(np.exp(logits - logits.max(1, keepdims=True)).sum(1, keepdims=True)) + logits.max(1, keepdims=True)
    logp = logits - lse
    t = np.full_like(logp, 0.2 / (logp.shape[1] - 1))
    t[np.arange(K), gold] = 0.8
    return dict(loss=np.array([-(t * logp).sum(), K], dtype=np.float64), logits=logits, feat6=feat6, group=group, gold=gold)


# ------------------------------------------------------------------------------------------------
# A.13  export  (model.py:525-605)   A.14  metrics  (model.py:608-655)
# ------------------------------------------------------------------------------------------------
def export_labels(part: Partition, layer: Layer, unmap, num_points, faithful=False):
    """-> (seg, ins, sem) int64 vectors in raw-vertex order."""
    seg = np.full(num_points, -1, dtype=np.int64)
    ins = np.full(num_points, -1, dtype=np.int64)
    sem = np.full(num_points, -1, dtype=np.int64)
    for c, m in enumerate(layer.members):
        r = part.root[layer.unmap[c]]
        seg[m] = part.root[r]
        if part.ins[r] != -1:
            ins[m] = part.ins[r] + 1
        if part.sem[r] != -1:
            sem[m] = part.sem[r] + 1
    unmap = np.asarray(unmap, dtype=np.int64)
    return seg[unmap], ins[unmap], sem[unmap]


def format_label_lines(vec, faithful=False):
    """The text payload of one layer_*.txt / final.*.txt file (model.py:536-547)."""
    if faithful:
        return "".join(["%d\n" % v for v in vec.tolist()])
    return "\n".join(map(str, vec.tolist())) + ("\n" if len(vec) else "")


def _accuracy(a, b):
    if a.shape[0] == 0:
        return float("nan")                       # sklearn: mean of empty -> nan (with a warning)
    return float(np.mean(a == b))


def evaluate(gt, sem_pred, ins_pred):
    """-> IoU_sem [1,2,40] f32, IoU_ins [1,2,40] f32, acc [4] f32."""
    gt = np.asarray(gt, dtype=np.int64)
    valid = gt[:, 0] != 0
    st, it = gt[valid, 0], gt[valid, 1]
    sp, ip = np.asarray(sem_pred)[valid], np.asarray(ins_pred)[valid]
    iou_sem = np.zeros((1, 2, 40), dtype=F32)
    for c in range(40):
        iou_sem[0, 0, c] = np.sum((sp == c + 1) & (st == c + 1))
        iou_sem[0, 1, c] = np.sum((sp == c + 1) | (st == c + 1))
    iou_ins = np.zeros((1, 2, 40), dtype=F32)
    for i in np.unique(ip):
        if i == -1:
            continue
        slot = int(sp[np.nonzero(ip == i)[0][0]]) - 1   # negative slot wraps (Python indexing)
        iou_ins[0, 0, slot] += np.sum((ip == i) & (it == i))
        iou_ins[0, 1, slot] += np.sum((ip == i) | (it == i))
    sv = np.isin(st, SEM_VALID_CLASS_IDS)
    iv = np.isin(it, INS_VALID_CLASS_IDS)            # sic: instance ids tested against class ids
    acc = np.array([_accuracy(st, sp), _accuracy(it, ip), _accuracy(st[sv], sp[sv]), _accuracy(it[iv], ip[iv])],
                   dtype=np.float64).astype(F32)
    return iou_sem, iou_ins, acc


# ------------------------------------------------------------------------------------------------
# SegModel.forward  (model.py:684-897)
# ------------------------------------------------------------------------------------------------
def forward_scene(scene, W, mode="ins_infer", faithful=False, keep=False):
    """Run the whole hot path on one scene (`seggroup_amd.synthetic.Scene`-like object with
    data / weak_label / seg / adj / unmap / gt).  Returns a dict with `labels` (name -> int64[V]),
    `metrics` (IoU_sem, IoU_ins, acc) and, if keep=True, stage intermediates."""
    data = np.asarray(scene.data, dtype=F32)
    N = data.shape[0]
    st = {}
    labels = {}

    def export(tag, layer):
        s, i, m = export_labels(part, layer, scene.unmap, N, faithful)
        if tag != "final":
            labels[f"{tag}.seg"] = s
        labels[f"{tag}.ins"] = i
        labels[f"{tag}.sem"] = m
        return i, m

    part = Partition(scene.weak_label[:, 1], scene.weak_label[:, 0], scene.seg)
    L1 = Layer(part)
    adj1 = contract_edges(scene.adj, part, np.arange(N), L1, faithful)
    export("layer_1", L1)

    # structural grouping layer (model.py:745-783)
    samples, _ = sample_clusters(data, L1, 64, transform=True)
    feat1 = mlp1_forward(samples, W)
    d1 = edge_distance(feat1, adj1)
    conn, stalled = group_nearby(part, d1, adj1, L1, 3 if mode == "sem_infer" else 6)
    L2 = Layer(part)
    g21, _ = L2.parents_of(L1, part)
    adj2 = contract_edges(adj1[~conn], part, L1.unmap, L2, faithful)
    feat2 = group_max(feat1, g21)
    ins_pred, sem_pred = export("layer_2", L2)
    if keep:
        st.update(samples=samples, feat1=feat1, d1=d1, adj1=adj1, adj2=adj2, root2=part.root.copy(), g21=g21)
    trace = [L1.count, L2.count]
    if mode == "sem_infer":
        return dict(labels=labels, metrics=evaluate(scene.gt, sem_pred, ins_pred), trace=trace, stages=st,
                    stalled=stalled)

    def semantic_layer(Lc, feat_c, adj_c, which, gcn_key):
        knn = cluster_knn(data[:, :3], Lc, 20)
        x9 = centre_per_cluster(data, Lc)
        pf = edgeconv_forward(x9, knn, W, which)
        fc = np.concatenate([feat_c, group_max(pf, Lc.members)], axis=1)
        fg = gcn_forward(fc, adj_c, W[gcn_key])
        d = edge_distance(fg, adj_c)
        conn, stl = group_nearby(part, d, adj_c, Lc, 2)
        Ln = Layer(part)
        groups, _ = Ln.parents_of(Lc, part)
        adj_n = contract_edges(adj_c[~conn], part, Lc.unmap, Ln, faithful)
        if keep:
            st[which] = dict(point_feat=pf, cat=fc, gcn=fg, d=d, adj=adj_n, root=part.root.copy(), knn=knn,
                             members=Lc.members, x9=x9, adj_in=np.asarray(adj_c, dtype=np.int64).reshape(-1, 2), groups=groups)
        return Ln, group_max(fg, groups), adj_n, stl

    L3, feat3, adj3, s2 = semantic_layer(L2, feat2, adj2, "mlp_2", "gcn_2.fc.weight")
    export("layer_3", L3)
    L4, feat4, adj4, s3 = semantic_layer(L3, feat3, adj3, "mlp_3", "gcn_3.fc.weight")
    export("layer_4", L4)
    groups45 = []
    feat5, adj5 = group_unlabeled(part, feat4, adj4, L4, data, faithful, record=groups45)
    L5 = Layer(part)
    ins_pred, sem_pred = export("final", L5)
    trace += [L3.count, L4.count, L5.count]
    if keep:
        st.update(feat4=feat4, adj4=adj4, feat5=feat5, adj5=adj5, root5=part.root.copy(), groups45=groups45,
                  ins5=np.array([part.ins[part.root[r]] for r in L5.unmap], dtype=np.int64),
                  sem5=np.array([part.sem[part.root[r]] for r in L5.unmap], dtype=np.int64))
    return dict(labels=labels, metrics=evaluate(scene.gt, sem_pred, ins_pred), trace=trace, stages=st,
                stalled=stalled or s2 or s3)
