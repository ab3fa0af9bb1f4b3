"""Module-level functions of the reference's `seggroup/model.py`, same names and argument meaning, on HIP.

The reference's `SegModel.forward` resolves these by global name, so they are the
operator seam a maintainer (or a test) can hook one at a time.  Each wrapper takes / returns torch
tensors like the reference and calls the C ABI (`include/seggroup_hip.h`); none of them has a CPU path.
`seggroup_amd.model.SegModel.forward` itself does NOT go through this file (it makes one
`sg_pipeline_forward` call); these wrappers exist for piecewise adoption and piecewise testing.

Deviations from the reference signatures (documented, deliberate):
  * `DisjointSet` is built with `DisjointSet.from_seg_lists(weak_ins, weak_sem, lists)` (the reference
    pokes `ds.indexs / cluster_id / point_num` by hand, model.py:712-721); read access to those attributes works.
  * `farthest_point_sampling` returns the indices only (the `[1,k,n]` distance cube the reference also
    returns is dead weight on its only call sites, model.py:406).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import hip

_i32 = torch.int32


def _need_cuda(t: torch.Tensor, what: str) -> torch.Tensor:
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{what} must be a CUDA/HIP tensor: seggroup_amd has no CPU path")
    return t


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def _csr(groups: Sequence[Sequence[int]]):
    off = np.zeros(len(groups) + 1, dtype=np.int32)
    np.cumsum([len(g) for g in groups], out=off[1:])
    flat = np.concatenate([np.asarray(g, dtype=np.int32) for g in groups]) if len(groups) else np.zeros(0, np.int32)
    return flat.astype(np.int32), off


def _tiles(off: np.ndarray, width: int = 256):
    tc, lo, hi, cto = [], [], [], [0]
    for c in range(len(off) - 1):
        for s in range(int(off[c]), int(off[c + 1]), width):
            tc.append(c); lo.append(s); hi.append(min(s + width, int(off[c + 1])))
        cto.append(len(tc))
    return [np.asarray(a, dtype=np.int32) for a in (tc, lo, hi, cto)]


# ------------------------------------------------------------------------------------------------
# DisjointSet (model.py:169-214) on top of the segment-level C++ engine
# ------------------------------------------------------------------------------------------------
class DisjointSet:
    """Point-keyed view of `sg_partition` (ids are point indices of cluster roots, like the reference)."""

    def __init__(self, weak_ins_label, weak_sem_label, seg=None):
        ins = np.asarray(weak_ins_label.cpu() if hasattr(weak_ins_label, "cpu") else weak_ins_label, dtype=np.int64)
        sem = np.asarray(weak_sem_label.cpu() if hasattr(weak_sem_label, "cpu") else weak_sem_label, dtype=np.int64)
        self.size = int(ins.shape[0])
        if seg is None:
            seg = np.arange(self.size, dtype=np.int32)           # singletons, like the reference constructor
        self._init(ins, sem, np.ascontiguousarray(seg, dtype=np.int32))

    @classmethod
    def from_seg_lists(cls, weak_ins_label, weak_sem_label, lists):
        from .scene import seg_from_lists
        n = len(lists)
        return cls(weak_ins_label, weak_sem_label, seg_from_lists(lists, n))

    def _init(self, ins, sem, seg):
        self._lib = hip.lib()
        self.seg = seg
        self.S = int(seg.max()) + 1 if seg.size else 0
        order = np.argsort(seg, kind="stable").astype(np.int32)
        counts = np.bincount(seg, minlength=self.S).astype(np.int32)
        off = np.zeros(self.S + 1, dtype=np.int32)
        np.cumsum(counts, out=off[1:])
        self.seg_points, self.seg_off = order, off
        self.seg_first = order[off[:-1]].astype(np.int32)
        self.seg_size = counts
        self._first_to_seg = {int(p): s for s, p in enumerate(self.seg_first)}
        si = np.ascontiguousarray(ins[self.seg_first], dtype=np.int32)
        ss = np.ascontiguousarray(sem[self.seg_first], dtype=np.int32)
        self._p = self._lib.sg_partition_create(self.S, self.seg_first.ctypes.data, self.seg_size.ctypes.data, si.ctypes.data, ss.ctypes.data)
        if not self._p:
            raise hip.SgError(hip.SG_EINVAL, self._lib.sg_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "_p", None):
                self._lib.sg_partition_destroy(self._p)
        except Exception:
            pass

    # -- reference API ---------------------------------------------------------------------------
    def _seg_of_root(self, pid) -> int:
        return self._first_to_seg[int(pid)]

    def find(self, idx):
        s = int(self.seg[int(idx)])
        return int(self.seg_first[self._lib.sg_partition_find(self._p, s)])

    def union(self, id1, id2):
        hip.check(self._lib.sg_partition_union(self._p, self._seg_of_root(id1), self._seg_of_root(id2)))

    def connected(self, idx1, idx2):
        return self.find(idx1) == self.find(idx2)

    def _label(self, idx):
        i, s, n = C.c_int32(), C.c_int32(), C.c_double()
        r = self._lib.sg_partition_find(self._p, int(self.seg[int(idx)]))
        hip.check(self._lib.sg_partition_label(self._p, r, C.byref(i), C.byref(s), C.byref(n)))
        return i.value, s.value, n.value

    def get_point_num(self, idx):
        return self._label(idx)[2]

    def get_weak_ins_label(self, idx):
        return np.int64(self._label(idx)[0])

    def get_weak_sem_label(self, idx):
        return np.int64(self._label(idx)[1])

    def get_cluster_id(self, idx):
        return np.int64(self.find(idx))

    def layer(self):
        S = self.S
        a = [np.zeros(S + 1, dtype=np.int32) for _ in range(6)]
        c = hip.check(self._lib.sg_partition_layer(self._p, *[x.ctypes.data for x in a]))
        root, cl_of_seg, order, cso, cpo, dst = a
        return dict(C=c, root=root[:c].copy(), cl_of_seg=cl_of_seg[:S].copy(), order=order[:S].copy(), cl_seg_off=cso[:c + 1].copy(),
                    cl_pt_off=cpo[:c + 1].copy(), dst=dst[:S].copy())

    def get_cluster_list(self):
        L = self.layer()
        out = []
        for c in range(L["C"]):
            segs = L["order"][L["cl_seg_off"][c]:L["cl_seg_off"][c + 1]]
            out.append(np.concatenate([self.seg_points[self.seg_off[s]:self.seg_off[s + 1]] for s in segs]).tolist())
        return out

    @property
    def cluster_id(self):
        L = self.layer()
        return self.seg_first[L["root"]][L["cl_of_seg"]][self.seg].astype(np.int64)

    @property
    def indexs(self):
        lists = [[] for _ in range(self.size)]
        for m in self.get_cluster_list():
            lists[self.find(m[0])] = m
        return lists


def _cluster_maps(ds: DisjointSet):
    """(cluster dict, cluster_map root->i, cluster_unmap i->root) of the re-index blocks (model.py:759-768)."""
    lists = ds.get_cluster_list()
    cluster = {i: m for i, m in enumerate(lists)}
    unmap = {i: ds.find(m[0]) for i, m in enumerate(lists)}
    return cluster, {r: i for i, r in unmap.items()}, unmap


# ------------------------------------------------------------------------------------------------
# graph bookkeeping
# ------------------------------------------------------------------------------------------------
def update_adj(adj_old, ds: DisjointSet, cluster_unmap_old: Dict[int, int], cluster_map_new: Dict[int, int]):
    """model.py:291-302.  `adj_old` [E,2] int64 indexes the OLD numbering (`cluster_unmap_old`: index -> root
    point id); returns sorted unique rows in the NEW numbering, or an empty 1-D tensor when nothing survives."""
    lib = hip.lib()
    adj = adj_old.reshape(-1, 2)
    E = int(adj.shape[0])
    if E == 0:
        return torch.LongTensor([])
    if adj.is_cuda and len(cluster_unmap_old) == ds.size and ds.layer()["C"] == ds.S:
        # first call (model.py:733): point-level mesh edges through the over-segmentation, on the device
        seg = torch.from_numpy(ds.seg).to(adj.device)
        out = torch.empty((E, 2), dtype=_i32, device=adj.device)
        cnt = torch.zeros(4, dtype=_i32, device=adj.device)
        ws = _ws(lib.sg_contract_ws_bytes(ds.S), adj.device)
        a64 = adj.contiguous().to(torch.int64)
        hip.check(lib.sg_contract_point_edges(a64.data_ptr(), E, seg.data_ptr(), ds.size, ds.S, out.data_ptr(), E, cnt.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _stream()))
        n = int(cnt[0].item())
        res = out[:n].to(torch.int64)
        # the engine numbers clusters by ascending root, which is what cluster_map_new encodes for a fresh partition
        return res if n else torch.LongTensor([])
    a = adj.detach().cpu().numpy().astype(np.int64)
    root_old = np.array([ds._seg_of_root(cluster_unmap_old[i]) for i in range(len(cluster_unmap_old))], dtype=np.int32)
    a32 = np.ascontiguousarray(a, dtype=np.int32)
    out = np.zeros((E, 2), dtype=np.int32)
    n = hip.check(lib.sg_partition_contract(ds._p, root_old.ctypes.data, a32.ctypes.data, E, None, out.ctypes.data))
    if n == 0:
        return torch.LongTensor([])
    return torch.from_numpy(out[:n].astype(np.int64))


def group_nearby_clusters(ds: DisjointSet, Dist, adj, group_unmap: Dict[int, int], th):
    """model.py:218-258 -> (ds, adj_connected, adj_unconnected)."""
    lib = hip.lib()
    a = adj.detach().cpu().numpy().reshape(-1, 2)
    E = a.shape[0]
    root = np.array([ds._seg_of_root(group_unmap[i]) for i in range(len(group_unmap))], dtype=np.int32)
    d = np.ascontiguousarray(Dist.detach().float().cpu().numpy(), dtype=np.float32)
    a32 = np.ascontiguousarray(a, dtype=np.int32)
    conn = np.zeros(max(E, 1), dtype=np.uint8)
    rc = lib.sg_partition_group_nearby(ds._p, root.ctypes.data, len(root), d.ctypes.data, a32.ctypes.data, E, C.c_float(float(th)),
                                       conn.ctypes.data)
    if rc != hip.SG_ESTALL:
        hip.check(rc)
    conn = conn[:E].astype(bool)
    pick = lambda m: adj[torch.from_numpy(np.nonzero(m)[0])] if m.any() else torch.LongTensor([])
    return ds, pick(conn), pick(~conn)


def calculate_distance(Feat, adj):
    """model.py:269-274."""
    _need_cuda(Feat, "Feat")
    lib = hip.lib()
    f = Feat.contiguous().float()
    a = adj.to(Feat.device).reshape(-1, 2).to(_i32).contiguous()
    out = torch.empty(a.shape[0], dtype=torch.float32, device=Feat.device)
    hip.check(lib.sg_edge_distance(f.data_ptr(), f.shape[1], f.shape[1], a.data_ptr(), a.shape[0], out.data_ptr(), _stream()))
    return out


def calculate_similarity(Feat, adj, alpha=1):
    """model.py:262-265."""
    return torch.exp(-calculate_distance(Feat, adj) * alpha)


def build_similarity_matrix(sims, adj, size):
    """model.py:305-309 (dense; the pipeline never builds it -- sg_gcn_forward works on the sparse graph)."""
    m = torch.eye(size, device=sims.device)
    m[adj[:, 0], adj[:, 1]] = sims
    m[adj[:, 1], adj[:, 0]] = sims
    return m


def build_distance_matrix(dists, adj, size):
    """model.py:312-316."""
    m = torch.ones(size, size, device=dists.device) * 1000
    m[adj[:, 0], adj[:, 1]] = dists
    m[adj[:, 1], adj[:, 0]] = dists
    return m


def aggregate_cluster_feature(Feat_old, clusters_new: Dict[int, List[int]], use_avg=False):
    """model.py:278-288: element-wise max over each group's rows."""
    _need_cuda(Feat_old, "Feat_old")
    lib = hip.lib()
    f = Feat_old.contiguous().float()
    gidx, goff = _csr([clusters_new[i] for i in range(len(clusters_new))])
    dev = f.device
    D = int(f.shape[1])
    d_off, d_idx = torch.from_numpy(goff).to(dev), torch.from_numpy(gidx).to(dev)
    # use_avg=True (model.py:282-284; never on the reference's own forward): rows of [max | mean], 2 D wide
    out = torch.empty((len(clusters_new), 2 * D if use_avg else D), dtype=torch.float32, device=dev)
    hip.check(lib.sg_group_max_rows(f.data_ptr(), D, D, d_off.data_ptr(), d_idx.data_ptr(), len(clusters_new), out.data_ptr(), out.shape[1], _stream()))
    if use_avg:
        hip.check(lib.sg_group_mean_rows(f.data_ptr(), D, D, d_off.data_ptr(), d_idx.data_ptr(), len(clusters_new), out.data_ptr() + 4 * D, out.shape[1],
                                         _stream()))
    return out


# ------------------------------------------------------------------------------------------------
# sampling, kNN, centring
# ------------------------------------------------------------------------------------------------
def l2_norm(x, y):
    """model.py:319-326."""
    return ((x - y) ** 2).sum(axis=2)


def _members_on(ds_or_lists, dev):
    lists = ds_or_lists.get_cluster_list() if isinstance(ds_or_lists, DisjointSet) else [ds_or_lists[i] for i in range(len(ds_or_lists))]
    members, off = _csr(lists)
    return lists, members, off, torch.from_numpy(members).to(dev), torch.from_numpy(off).to(dev)


def farthest_point_sampling(pts, k, initial_idx=None, metrics=l2_norm, skip_initial=False, indices_dtype=np.int32,
                            distances_dtype=np.float32):
    """model.py:329-395 with the reference's signature and defaults: `pts` [n,dim] or [B,n,dim] CUDA tensor -> (indices [B,k], distances
    [B,k,n]) as NumPy arrays like the reference (one workgroup per cloud, `sg_fps_general`: l2_norm in NumPy's order, first-index argmax).
    initial_idx=None draws the start like the reference (np.random.randint); `metrics` other than l2_norm is refused: the kernel IS l2_norm.
    The forward itself never comes here: its sampling (initial_idx=0, skip_initial=True AND the trailing-zero fix-up of
    get_cluster_pointcloud, model.py:398-426) is the hot path's `sg_fps_sample`."""
    if metrics is not l2_norm:
        raise ValueError("farthest_point_sampling: only metrics=l2_norm exists on the device")
    _need_cuda(pts, "pts")
    lib = hip.lib()
    p3 = (pts if pts.dim() == 3 else pts[None]).contiguous().float()
    B, n, dim = (int(v) for v in p3.shape)
    if k <= 0:
        raise ValueError("farthest_point_sampling: k must be positive")
    idx = torch.empty((B, k), dtype=_i32, device=p3.device)
    dist = torch.empty((B, k, n), dtype=torch.float32, device=p3.device)
    ws = _ws(lib.sg_fps_general_ws_bytes(n), p3.device)
    start = int(np.random.randint(n)) if initial_idx is None else int(initial_idx)      # one draw for the whole batch (model.py:374)
    for b in range(B):
        hip.check(lib.sg_fps_general(p3[b].data_ptr(), n, dim, k, start, int(bool(skip_initial)), idx[b].data_ptr(), dist[b].data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream()))
    return idx.cpu().numpy().astype(indices_dtype), dist.cpu().numpy().astype(distances_dtype)


def get_cluster_pointcloud(data, ds: DisjointSet, point_num=128, transfrom=True):
    """model.py:398-426 -> [S, point_num, C] (C = data.shape[1])."""
    _need_cuda(data, "data")
    lib = hip.lib()
    dev = data.device
    d = data.contiguous().float()
    lists, members, off, d_m, d_o = _members_on(ds, dev)
    Cn, ch = len(lists), int(d.shape[1])
    out = torch.empty((Cn, point_num, ch), dtype=torch.float32, device=dev)
    ws = _ws(lib.sg_fps_ws_bytes(d.shape[0]), dev)
    hip.check(lib.sg_fps_sample(d.data_ptr(), d.shape[0], ch, d_m.data_ptr(), d_o.data_ptr(), Cn, point_num, ch, int(bool(transfrom)),
                                out.data_ptr(), None, ws.data_ptr(), ws.numel(), _stream()))
    return out


def _center(data6, lists, dev):
    lib = hip.lib()
    members, off = _csr(lists)
    tc, lo, hi, cto = _tiles(off)
    N = int(data6.shape[0])
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(m=members, o=off, tc=tc, lo=lo, hi=hi, cto=cto).items()}
    x9m = torch.empty((N, 12), dtype=torch.float32, device=dev)
    xyzw = torch.empty((N, 4), dtype=torch.float32, device=dev)
    ws = _ws(lib.sg_center_ws_bytes(len(tc), len(lists)), dev)
    hip.check(lib.sg_center_clusters(data6.data_ptr(), N, t["m"].data_ptr(), t["o"].data_ptr(), len(lists), t["tc"].data_ptr(),
                                     t["lo"].data_ptr(), t["hi"].data_ptr(), len(tc), t["cto"].data_ptr(), x9m.data_ptr(), xyzw.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _stream()))
    return members, off, t, x9m, xyzw


def combine_centralized_pointcloud(data, ds: DisjointSet):
    """model.py:429-436 -> [N,9] in point order."""
    _need_cuda(data, "data")
    d = data.contiguous().float()
    members, _, t, x9m, _ = _center(d, ds.get_cluster_list(), d.device)
    out = torch.empty((d.shape[0], 9), dtype=torch.float32, device=d.device)
    out[t["m"].long()] = x9m[:, :9]
    return out


def get_knn(data, cluster: Dict[int, List[int]], k=20):
    """model.py:512-522: in-cluster kNN table [N,k] int64 (point ids, point order), `data` = XYZ [N,3]."""
    _need_cuda(data, "data")
    lib = hip.lib()
    dev = data.device
    N = int(data.shape[0])
    d6 = torch.zeros((N, 6), dtype=torch.float32, device=dev)
    d6[:, :3] = data[:, :3]
    lists = [cluster[i] for i in range(len(cluster))]
    members, off, t, _, xyzw = _center(d6, lists, dev)
    pos_of_point = np.empty(N, dtype=np.int64)
    pos_of_point[members] = np.arange(members.shape[0])
    knn = torch.empty((N, k), dtype=_i32, device=dev)
    hip.check(lib.sg_cluster_knn(xyzw.data_ptr(), N, t["o"].data_ptr(), t["tc"].data_ptr(), t["lo"].data_ptr(), t["hi"].data_ptr(),
                                 int(t["tc"].shape[0]), k, int(pos_of_point[0]), knn.data_ptr(), _stream()))
    m = t["m"].long()
    out = torch.zeros((N, k), dtype=torch.int64, device=dev)
    out[m] = m[knn.long()]
    return out


def knn(x, k):
    """model.py:30-36: x [B,C,n] (C = 3) -> idx [B,n,k] int64, descending score, lower index first on ties."""
    _need_cuda(x, "x")
    B, Cc, n = x.shape
    if Cc != 3 or k != 20:
        # any channel count / any k <= min(n, 128) (model.py:30-36 as written; the forward itself only asks for C = 3 with k = 20 and, inside
        # MLP1, k = 10): the plain kernel of csrc/kernels_general.hip
        xc = x.contiguous().float()
        out = torch.empty((B, n, k), dtype=torch.int64, device=x.device)
        hip.check(hip.lib().sg_knn_general(xc.data_ptr(), int(B), int(Cc), int(n), int(k), out.data_ptr(), _stream()))
        return out
    pts = x.transpose(2, 1).reshape(B * n, 3)
    table = get_knn(pts, {b: list(range(b * n, (b + 1) * n)) for b in range(B)}, k)
    return (table.view(B, n, k) - (torch.arange(B, device=x.device) * n).view(B, 1, 1))


# ------------------------------------------------------------------------------------------------
# network blocks
# ------------------------------------------------------------------------------------------------
def mlp1_forward(x, conv_w, bn_w, bn_b):
    """MLP1.forward (model.py:73-80): x [S,6,64] -> [S,128]."""
    _need_cuda(x, "x")
    lib = hip.lib()
    s = x.transpose(2, 1).contiguous().float()
    S = int(s.shape[0])
    out = torch.empty((S, 128), dtype=torch.float32, device=x.device)
    ws = _ws(lib.sg_mlp1_ws_bytes(S), x.device)
    w = conv_w.reshape(64, 6).contiguous().float()
    hip.check(lib.sg_mlp1_forward(s.data_ptr(), S, w.data_ptr(), bn_w.contiguous().data_ptr(), bn_b.contiguous().data_ptr(), out.data_ptr(), 128,
                                  ws.data_ptr(), ws.numel(), _stream()))
    return out


def edgeconv_forward(x, idx, w1, g1, b1, w2=None, g2=None, b2=None):
    """MLP2.forward / MLP3.forward (model.py:114-138): x [1,9,N], idx [1,N,k] point ids -> [1,64,N]."""
    _need_cuda(x, "x")
    lib = hip.lib()
    N, k = int(x.shape[2]), int(idx.shape[-1])
    x12 = torch.zeros((N, 12), dtype=torch.float32, device=x.device)
    x12[:, :9] = x[0].transpose(1, 0)
    knn_ = idx.reshape(N, k).to(_i32).contiguous()
    out = torch.empty((N, 64), dtype=torch.float32, device=x.device)
    ws = _ws(lib.sg_edgeconv_ws_bytes(N), x.device)
    p = lambda t: None if t is None else t.reshape(t.shape[0], -1).contiguous().float().data_ptr()
    keep = [t.reshape(t.shape[0], -1).contiguous().float() if t is not None else None for t in (w1, g1, b1, w2, g2, b2)]
    hip.check(lib.sg_edgeconv_forward(x12.data_ptr(), knn_.data_ptr(), N, k, 1 if w2 is None else 2,
                                      *[None if t is None else t.data_ptr() for t in keep], out.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _stream()))
    return out.transpose(1, 0).unsqueeze(0)


def gcn_forward(X, Edge_adj, fc_weight, alpha=1 / 8):
    """calculate_similarity + build_similarity_matrix + GCN.forward (model.py:797-799,146-151) without the dense
    matrix: X [S,D], Edge_adj [E,2] (sorted unique rows) -> relu(fc(rownorm(I + sym(exp(-alpha d))) @ X))."""
    _need_cuda(X, "X")
    lib = hip.lib()
    S, D = int(X.shape[0]), int(X.shape[1])
    a = Edge_adj.detach().cpu().numpy().reshape(-1, 2).astype(np.int32)
    E = a.shape[0]
    rowptr = np.zeros(S + 1, dtype=np.int32)
    np.add.at(rowptr, a[:, 0] + 1, 1)
    np.add.at(rowptr, a[:, 1] + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    fill = rowptr[:-1].copy()
    col = np.zeros(2 * E, dtype=np.int32)
    eid = np.zeros(2 * E, dtype=np.int32)
    for e, (u, v) in enumerate(a):
        col[fill[u]] = v; eid[fill[u]] = e; fill[u] += 1
        col[fill[v]] = u; eid[fill[v]] = e; fill[v] += 1
    dev = X.device
    t = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (a, rowptr, col, eid)]
    x = X.contiguous().float()
    w = fc_weight.contiguous().float()
    out = torch.empty((S, D), dtype=torch.float32, device=dev)
    ws = _ws(lib.sg_gcn_ws_bytes(S, D, E), dev)
    hip.check(lib.sg_gcn_forward(x.data_ptr(), S, D, t[0].data_ptr(), E, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), w.data_ptr(),
                                 C.c_float(alpha), out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def _sym_csr(a: np.ndarray, S: int):
    E = a.shape[0]
    rowptr = np.zeros(S + 1, dtype=np.int32)
    np.add.at(rowptr, a[:, 0] + 1, 1)
    np.add.at(rowptr, a[:, 1] + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    fill = rowptr[:-1].copy()
    col = np.zeros(2 * E, dtype=np.int32)
    eid = np.zeros(2 * E, dtype=np.int32)
    for e, (u, v) in enumerate(a):
        col[fill[u]] = v; eid[fill[u]] = e; fill[u] += 1
        col[fill[v]] = u; eid[fill[v]] = e; fill[v] += 1
    return rowptr, col, eid


# ------------------------------------------------------------------------------------------------
# training step, operator level: backward of every operator of the chain + the classifier tail
# ------------------------------------------------------------------------------------------------
def gcn_backward(X, Edge_adj, fc_weight, grad_out, alpha=1 / 8):
    """Gradients of `gcn_forward` w.r.t. X and fc_weight, including the path through the similarity weights
    (model.py:262-265,305-309 are differentiated by autograd in the reference)."""
    _need_cuda(X, "X")
    lib = hip.lib()
    S, D = int(X.shape[0]), int(X.shape[1])
    a = Edge_adj.detach().cpu().numpy().reshape(-1, 2).astype(np.int32)
    E = a.shape[0]
    dev = X.device
    t = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (a,) + _sym_csr(a, S)]
    x, w, g = X.contiguous().float(), fc_weight.contiguous().float(), grad_out.contiguous().float()
    gx = torch.empty((S, D), dtype=torch.float32, device=dev)
    gw = torch.empty((D, D), dtype=torch.float32, device=dev)
    ws = _ws(lib.sg_gcn_backward_ws_bytes(S, D, E), dev)
    hip.check(lib.sg_gcn_backward(x.data_ptr(), S, D, t[0].data_ptr(), E, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), w.data_ptr(),
                                  C.c_float(alpha), g.data_ptr(), gx.data_ptr(), gw.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return gx, gw


def aggregate_cluster_feature_backward(Feat_old, clusters_new: Dict[int, List[int]], grad_new):
    """Gradient of `aggregate_cluster_feature` (max, model.py:278-288) w.r.t. Feat_old: a group's gradient goes to its first
    maximal row."""
    _need_cuda(Feat_old, "Feat_old")
    lib = hip.lib()
    dev = Feat_old.device
    G = len(clusters_new)
    gidx, goff = _csr([clusters_new[i] for i in range(G)])
    d_gidx, d_goff = torch.from_numpy(gidx).to(dev), torch.from_numpy(goff).to(dev)
    rows, g = Feat_old.contiguous().float(), grad_new.contiguous().float()
    D = int(rows.shape[1])
    out = torch.zeros_like(rows)
    hip.check(lib.sg_group_max_rows_backward(rows.data_ptr(), D, D, d_goff.data_ptr(), d_gidx.data_ptr(), G, g.data_ptr(), D, out.data_ptr(), D,
                                             _stream()))
    return out


def segment_max_backward(rows, cl_off, grad_out):
    """Gradient of the point -> cluster max (model.py:793,834): rows [N,D] in member order, cluster c = rows
    [cl_off[c], cl_off[c+1]); grad_out [C,D]."""
    _need_cuda(rows, "rows")
    lib = hip.lib()
    dev = rows.device
    r, g = rows.contiguous().float(), grad_out.contiguous().float()
    off = torch.as_tensor(np.asarray(cl_off, dtype=np.int32)).to(dev)
    N, D, Cn = int(r.shape[0]), int(r.shape[1]), int(off.shape[0]) - 1
    out = torch.zeros_like(r)
    ws = _ws(lib.sg_segment_max_backward_ws_bytes(Cn, D), dev)
    hip.check(lib.sg_segment_max_backward(r.data_ptr(), N, D, off.data_ptr(), Cn, g.data_ptr(), D, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def mlp1_backward(x, grad_feat, conv_w, bn_w, bn_b):
    """Parameter gradients of `mlp1_forward` (model.py:39-80): x [S,6,64] samples, grad_feat [S,128] ->
    dict(w [64,6], g [64], b [64], bn_stats [128])."""
    _need_cuda(x, "x")
    lib = hip.lib()
    dev = x.device
    S = int(x.shape[0])
    samples = x.transpose(2, 1).contiguous().float()                   # [S,64,6]
    g = grad_feat.contiguous().float()
    w, ga, be = conv_w.reshape(64, 6).contiguous().float(), bn_w.contiguous().float(), bn_b.contiguous().float()
    out = {"w": torch.empty((64, 6), device=dev), "g": torch.empty(64, device=dev), "b": torch.empty(64, device=dev),
           "bn_stats": torch.zeros(128, device=dev)}
    ws = _ws(lib.sg_mlp1_backward_ws_bytes(S), dev)
    hip.check(lib.sg_mlp1_backward(samples.data_ptr(), S, w.data_ptr(), ga.data_ptr(), be.data_ptr(), g.data_ptr(), 128, out["w"].data_ptr(),
                                   out["g"].data_ptr(), out["b"].data_ptr(), out["bn_stats"].data_ptr(), ws.data_ptr(), ws.numel(), _stream()))
    return out


def edgeconv_backward(x, idx, grad_out, w1, g1, b1, w2=None, g2=None, b2=None, bn2_stats=None):
    """Parameter gradients of `edgeconv_forward` (MLP2 / MLP3 with batch-statistics BatchNorm2d, model.py:83-138):
    x [1,9,N], idx [1,N,k], grad_out [1,64,N] -> dict(w1 [64,18], g1, b1[, w2 [64,64], g2, b2], bn_stats [256]).
    bn2_stats [128] (MLP3 only): the second BatchNorm's batch mean | variance from the forward; given, no dense forward pass is made."""
    _need_cuda(x, "x")
    lib = hip.lib()
    dev = x.device
    N, k = int(x.shape[2]), int(idx.shape[-1])
    x12 = torch.zeros((N, 12), dtype=torch.float32, device=dev)
    x12[:, :9] = x[0].transpose(1, 0)
    knn_ = idx.reshape(N, k).to(_i32).contiguous()
    go = grad_out[0].transpose(1, 0).contiguous().float()
    two = w2 is not None
    keep = [t.reshape(t.shape[0], -1).contiguous().float() if t is not None else None for t in (w1, g1, b1, w2, g2, b2)]
    out = {"w1": torch.empty((64, 18), device=dev), "g1": torch.empty(64, device=dev), "b1": torch.empty(64, device=dev),
           "bn_stats": torch.zeros(256, device=dev)}
    if two:
        out.update(w2=torch.empty((64, 64), device=dev), g2=torch.empty(64, device=dev), b2=torch.empty(64, device=dev))
    if bn2_stats is not None:
        bn2_stats = bn2_stats.contiguous().float()
    ws = _ws(lib.sg_edgeconv_backward_ws_bytes(N), dev)
    hip.check(lib.sg_edgeconv_backward(x12.data_ptr(), knn_.data_ptr(), N, k, 2 if two else 1, *[None if t is None else t.data_ptr() for t in keep],
                                       go.data_ptr(), out["w1"].data_ptr(), out["g1"].data_ptr(), out["b1"].data_ptr(),
                                       out["w2"].data_ptr() if two else None, out["g2"].data_ptr() if two else None,
                                       out["b2"].data_ptr() if two else None, hip.ptr(bn2_stats), out["bn_stats"].data_ptr(), ws.data_ptr(), ws.numel(),
                                       _stream()))
    return out


class TrainTail:
    """model.py:900-932: per-instance max feature -> Classifier (154-166) -> label-smoothed cross entropy (util.py:12-29),
    forward and backward on HIP.  `keep` is the pinned dropout mask [K,128] already scaled by 1 / (1 - p) (None = no dropout)."""

    def __init__(self, feat5, ins5, sem5, classifier: Dict[str, torch.Tensor], keep=None):
        _need_cuda(feat5, "feat5")
        self.lib = hip.lib()
        dev = feat5.device
        ins5 = np.asarray(ins5, dtype=np.int64)
        ins_gt = np.unique(ins5)                                       # model.py:909
        group = np.searchsorted(ins_gt, ins5).astype(np.int32)
        self.K, self.C = int(ins_gt.shape[0]), int(ins5.shape[0])
        gold = np.array([np.asarray(sem5)[np.nonzero(group == k)[0][0]] for k in range(self.K)], dtype=np.int32)
        if gold.min() < 0 or gold.max() >= 40:
            raise ValueError("TrainTail: a weak semantic label outside 0..39 (scatter in cross_entropy_loss would raise in the reference)")
        self.feat5 = feat5.contiguous().float()
        self.group, self.gold = torch.from_numpy(group).to(dev), torch.from_numpy(gold).to(dev)
        self.keep = None if keep is None else torch.as_tensor(np.asarray(keep, dtype=np.float32)).to(dev).contiguous()
        self.w = {k: classifier[k].detach().to(dev).contiguous().float() for k in
                  ("linear1.weight", "bn1.weight", "bn1.bias", "linear2.weight", "linear2.bias")}
        self.cls = hip.Classifier(w1=self.w["linear1.weight"].data_ptr(), gamma=self.w["bn1.weight"].data_ptr(), beta=self.w["bn1.bias"].data_ptr(),
                                  w2=self.w["linear2.weight"].data_ptr(), b2=self.w["linear2.bias"].data_ptr())
        self.ws = _ws(self.lib.sg_train_tail_ws_bytes(self.C, self.K), dev)
        self.logits = torch.empty((self.K, 40), dtype=torch.float32, device=dev)
        self.loss = torch.empty(2, dtype=torch.float32, device=dev)

    def forward(self):
        """-> loss [1,2] = [[loss_sum, K]] like the reference's return value"""
        hip.check(self.lib.sg_train_tail_forward(self.feat5.data_ptr(), self.C, self.group.data_ptr(), self.K, self.gold.data_ptr(),
                                                 hip.ptr(self.keep), C.byref(self.cls), self.logits.data_ptr(), self.loss.data_ptr(),
                                                 self.ws.data_ptr(), self.ws.numel(), _stream()))
        return self.loss.view(1, 2)

    def backward(self, scale: float = None):
        """Gradients of scale * loss_sum (default scale = 1 / K: train.py:164-166 on one GPU) -> dict of the five classifier
        gradients + 'feat5'."""
        dev = self.feat5.device
        if scale is None:
            scale = 1.0 / self.K
        g = {"linear1.weight": torch.empty((128, 256), device=dev), "bn1.weight": torch.empty(128, device=dev), "bn1.bias": torch.empty(128, device=dev),
             "linear2.weight": torch.empty((40, 128), device=dev), "linear2.bias": torch.empty(40, device=dev),
             "feat5": torch.empty((self.C, 256), device=dev)}
        hip.check(self.lib.sg_train_tail_backward(self.C, self.K, self.gold.data_ptr(), hip.ptr(self.keep), C.byref(self.cls), C.c_float(scale),
                                                  g["linear1.weight"].data_ptr(), g["bn1.weight"].data_ptr(), g["bn1.bias"].data_ptr(),
                                                  g["linear2.weight"].data_ptr(), g["linear2.bias"].data_ptr(), g["feat5"].data_ptr(),
                                                  self.ws.data_ptr(), self.ws.numel(), _stream()))
        return g


# ------------------------------------------------------------------------------------------------
# export + evaluate
# ------------------------------------------------------------------------------------------------
def _export(ds: DisjointSet, unmap, which: int, output_path: str):
    lib = hip.lib()
    tabs = [np.zeros(ds.S, dtype=np.int32) for _ in range(3)]
    hip.check(lib.sg_partition_export_tables(ds._p, *[t.ctypes.data for t in tabs]))
    um = unmap if isinstance(unmap, torch.Tensor) else torch.load(unmap)
    dev = torch.device("cuda", torch.cuda.current_device())
    d_un = um.to(dev).to(_i32).contiguous()
    d_seg = torch.from_numpy(ds.seg).to(dev)
    d_tab = torch.from_numpy(tabs[which]).to(dev)
    V = int(d_un.shape[0])
    out = torch.empty((1, V), dtype=_i32, device=dev)
    hip.check(lib.sg_export_labels(d_un.data_ptr(), V, d_seg.data_ptr(), ds.size, d_tab.data_ptr(), 1, ds.S, out.data_ptr(), _stream()))
    vec = out[0].cpu().numpy()
    if output_path:
        os.makedirs(os.path.dirname(output_path), exist_ok=True)
        hip.check(lib.sg_write_label_txt(output_path.encode(), vec.ctypes.data, V))
    return torch.from_numpy(vec.astype(np.int64))


def _label_path(output_root, layer, kind):
    name = f"final.{kind}.txt" if layer == "final" else f"layer_{int(layer)}.{kind}.txt"
    return os.path.join(output_root, name)


def export_segment_label(ds, ds_unmap, output_root, unmap_path, layer, point_num=150000):
    """model.py:525-549."""
    return _export(ds, unmap_path, 0, _label_path(output_root, layer, "seg"))


def export_instance_label(ds, ds_unmap, output_root, unmap_path, layer, point_num=150000):
    """model.py:552-577."""
    return _export(ds, unmap_path, 1, _label_path(output_root, layer, "ins"))


def export_semantic_label(ds, ds_unmap, output_root, unmap_path, layer, point_num=150000):
    """model.py:580-605."""
    return _export(ds, unmap_path, 2, _label_path(output_root, layer, "sem"))


def evaluate(scene_name, sem_pred, ins_pred, root="."):
    """model.py:608-655 -> (IoU_sem [1,2,40], IoU_ins [1,2,40], acc [4]) float32 CPU tensors."""
    lib = hip.lib()
    gt = torch.load(os.path.join(root, "dataset", "scannet", "label", "real", "raw", scene_name, scene_name + ".label.pth"))
    dev = torch.device("cuda", torch.cuda.current_device())
    d_gt = gt.to(dev).to(_i32).contiguous()
    sp = sem_pred.to(dev).to(_i32).contiguous()
    ip = ins_pred.to(dev).to(_i32).contiguous()
    V = int(sp.shape[0])
    max_ins = int(max(int(ip.max().item()), 0)) + 2
    a, b, c = np.zeros(80, np.float32), np.zeros(80, np.float32), np.zeros(4, np.float32)
    ws = _ws(lib.sg_eval_ws_bytes(max_ins), dev)
    hip.check(lib.sg_evaluate(d_gt.data_ptr(), sp.data_ptr(), ip.data_ptr(), V, max_ins, a.ctypes.data, b.ctypes.data, c.ctypes.data,
                              ws.data_ptr(), ws.numel(), _stream()))
    return torch.from_numpy(a.reshape(1, 2, 40)), torch.from_numpy(b.reshape(1, 2, 40)), torch.from_numpy(c)
