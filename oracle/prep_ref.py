"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's pre-processing math.

NumPy restatement of the compute parts of `seggroup/dataset/scannet/util.py` that turn a raw ScanNet scan into
the hot path's inputs.  Pinned against outputs of the REAL reference functions run in the build container
(`tools/capture_prepare.py` -> `tests/golden/prep_*.npz`); only `tests/` may import this module.

    sample_points      generate_pointcloud_pth   util.py:633-693   (colour centring, mapper, unmapper)
    get_unmapper       get_unmapper              util.py:538-550   (+ cal_pairwise_distance 530-535)
    get_adj_from_pointcloud  get_adj_from_pointcloud  util.py:814-834
    get_adj_from_mesh  get_adj_from_mesh         util.py:771-792
    segment_lists      generate_seg_labels_and_ds_set  util.py:174-220
"""
from __future__ import annotations

import json

import numpy as np

F32 = np.float32


def centre_colours(rgb_u8):
    """util.py:655 in float64 (`np.zeros([V,6])`), then `torch.FloatTensor` rounds to fp32 (663)."""
    return (np.asarray(rgb_u8, dtype=np.float64) / 127.5 - 1).astype(F32)


def make_mapper(num_vertices: int, num_points: int, perm):
    """util.py:664-676: `arange(V).repeat(num_points // V)` followed by the first `num_points % V` entries of a random
    permutation (`perm` stands for the `torch.randperm(V)` draw)."""
    rep, rem = num_points // num_vertices, num_points % num_vertices
    parts = [np.tile(np.arange(num_vertices, dtype=np.int64), rep)] if rep else []
    parts.append(np.asarray(perm, dtype=np.int64)[:rem])
    return np.concatenate(parts)


def pairwise_scores(x, y):
    """util.py:530-535 in fp32:  s[i,j] = ((-xx_i) - inner_ij) - yy_j,  inner = -2 * (x @ y.T).
    The K=3 dot product is MKL's fma(z,z', fma(y,y', fl(x*x'))) (same convention as oracle/cpu_ref.knn_scores, there
    pinned by the kNN tables of the model captures), squares are rounded separately and summed left to right."""
    import torch
    q = torch.from_numpy(np.ascontiguousarray(x, dtype=F32))
    a = torch.from_numpy(np.ascontiguousarray(y, dtype=F32))
    qd, ad = q.double(), a.double()
    t = q[:, 0:1] * a[:, 0][None, :]
    t = torch.addcmul(t.double(), qd[:, 1:2], ad[:, 1][None, :]).float()
    t = torch.addcmul(t.double(), qd[:, 2:3], ad[:, 2][None, :]).float()
    inner = t * -2.0

    def sq(v):
        s = v * v
        return (s[:, 0] + s[:, 1]) + s[:, 2]
    return (((-sq(q))[:, None] - inner) - sq(a)[None, :]).numpy()


def get_unmapper(x, y, chunk: int = 4096):
    """util.py:538-550: index of the best-scoring (nearest) row of y for every row of x.
    `topk(k=1)` leaves ties unspecified; the build DEFINES: the lowest index among equal best scores."""
    x = np.asarray(x, dtype=F32)
    out = np.empty(x.shape[0], dtype=np.int64)
    for i in range(0, x.shape[0], chunk):
        out[i:i + chunk] = np.argmax(pairwise_scores(x[i:i + chunk], y), axis=1)       # argmax = first maximum
    return out


def tie_rows(x, y, chunk: int = 4096):
    """Rows of x whose best score is attained by more than one row of y (where the reference's choice is unspecified)."""
    x = np.asarray(x, dtype=F32)
    bad = np.zeros(x.shape[0], dtype=bool)
    for i in range(0, x.shape[0], chunk):
        s = pairwise_scores(x[i:i + chunk], y)
        bad[i:i + chunk] = (s == s.max(axis=1, keepdims=True)).sum(axis=1) > 1
    return bad


def get_adj_from_pointcloud(points, k: int = 10, chunk: int = 2048):
    """util.py:814-834: every point's k best-scoring other rows of the cloud (topk(k + 1) with the top entry dropped) as
    per-row sorted, unique [*, 2] int64 rows.  `topk` leaves the order of equal scores unspecified; the build DEFINES: lower index
    first.  -> (adj, tie_rows) where tie_rows marks the points whose top k + 2 scores contain an equal pair (there the reference's
    own choice may differ)."""
    x = np.ascontiguousarray(np.asarray(points, dtype=F32)[:, :3])
    n = x.shape[0]
    pairs = np.empty((n, k, 2), dtype=np.int64)
    ties = np.zeros(n, dtype=bool)
    for i in range(0, n, chunk):
        s = pairwise_scores(x[i:i + chunk], x)
        order = np.argsort(-s, axis=1, kind="stable")[:, :k + 2]          # descending score, ascending index inside a tie
        top = np.take_along_axis(s, order, axis=1)
        ties[i:i + chunk] = (top[:, 1:] == top[:, :-1]).any(axis=1)
        pairs[i:i + chunk, :, 0] = np.arange(i, min(n, i + chunk))[:, None]
        pairs[i:i + chunk, :, 1] = order[:, 1:k + 1]
    return np.unique(np.sort(pairs.reshape(-1, 2), axis=1), axis=0), ties


def sample_points(xyz, rgb_u8, num_points: int, perm):
    """generate_pointcloud_pth (util.py:633-693) without the file I/O:
    -> pointcloud_sampled [num_points,6] f32, mapper [num_points] i64, unmapper [V] i64."""
    v = np.asarray(xyz).shape[0]
    cloud = np.concatenate([np.asarray(xyz, dtype=np.float64), np.asarray(rgb_u8, dtype=np.float64) / 127.5 - 1], 1).astype(F32)
    mapper = make_mapper(v, num_points, perm)
    sampled = cloud[mapper]
    unmapper = np.full(v, -100, dtype=np.int64)
    unmapper[mapper] = np.arange(mapper.shape[0])          # util.py:687-689: a later duplicate overwrites (NumPy: last write wins)
    missing = np.nonzero(unmapper == -100)[0]
    if missing.size:
        unmapper[missing] = get_unmapper(cloud[missing, :3], sampled[:, :3])
    return sampled, mapper, unmapper


def get_adj_from_mesh(faces, unmapper=None):
    """util.py:771-792: the three edges of every face, zero-length ones dropped (783) -> (raw, resampled) edge lists,
    each with ascending vertex ids per row and unique rows in lexicographic order (torch.sort(dim=-1), torch.unique(dim=0)).
    Edges that collapse only AFTER unmapping stay as (a, a) rows, as in the reference."""
    f = np.asarray(faces, dtype=np.int64)
    e = np.concatenate([f[:, [0, 1]], f[:, [0, 2]], f[:, [1, 2]]], 1).reshape(-1, 2)      # per face: (0,1), (0,2), (1,2)
    e = e[e[:, 0] != e[:, 1]]
    res = None
    if unmapper is not None:
        res = np.unique(np.sort(np.asarray(unmapper, dtype=np.int64)[e], axis=1), axis=0)
    return np.unique(np.sort(e, axis=1), axis=0), res


def segment_lists(seg_indices, mapper):
    """generate_seg_labels_and_ds_set (util.py:174-220) without the file I/O:
    -> raw labels compacted to 0..S-1 in ascending id order [V] i64 (the `.seg.txt` column), and the `.seg.json`
    payload: list i holds the ascending sampled-point indices of a segment iff i is its smallest member."""
    lab = np.asarray(seg_indices, dtype=np.int64)
    raw = np.searchsorted(np.unique(lab), lab)                 # seg_remapper[x] = rank of x (util.py:181-186)
    sampled = raw[np.asarray(mapper, dtype=np.int64)]
    lists = [[] for _ in range(sampled.shape[0])]
    order = np.argsort(sampled, kind="stable")
    bounds = np.nonzero(np.diff(sampled[order]))[0] + 1
    for grp in np.split(order, bounds):
        lists[int(grp[0])] = grp.tolist()
    return raw, lists


def seg_json_text(lists) -> str:
    return json.dumps(lists)                                   # util.py:218-219 (json.dump default separators)
