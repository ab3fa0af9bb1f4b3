"""Deterministic synthetic ScanNet-shaped scenes (SURVEY.md section 8d recipe).

A scene is everything the reference hot path reads for one ScanNet scan:

  data        [N,6]  f32   XYZ (metres, 8x6x3 box) + RGB in [-1,1]    (<scene>.pcl.pth,   seggroup/data.py:34)
  weak_label  [N,2]  i64   [sem 0..39|-1, ins 0..K-1|-1]              (<scene>.label.pth, seggroup/data.py:36)
  seg         [N]    i32   over-segment id per point, 0..S-1 in order of first point
                           (<scene>.seg.json lists, seggroup/dataset/scannet/util.py:205-220)
  adj         [E0,2] i64   point adjacency, each row sorted, rows lexicographically unique
                           (<scene>.adj.pth, seggroup/dataset/scannet/util.py:771-792)
  unmap       [V]    i64   raw vertex -> resampled point                (<scene>.unmap.pth, util.py:687-693)
  gt          [V,2]  i64   [sem+1, ins+1] per raw vertex, 0 = unannotated (label/real/raw, util.py:697-729)

The generator is NumPy + SciPy only and uses a counter-based splitmix64 stream, so the
GPU box regenerates byte-identical inputs without any torch RNG state.
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Optional

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, stream: int, n: int) -> np.ndarray:
    """n 64-bit words of the splitmix64 sequence keyed by (seed, stream); counter-based."""
    with np.errstate(over="ignore"):
        base = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)) ^ (np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, stream: int, n: int) -> np.ndarray:
    """float32 uniforms in [0,1) with 24 random bits each."""
    return ((splitmix64(seed, stream, n) >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def randint(seed: int, stream: int, n: int, hi: int) -> np.ndarray:
    return (splitmix64(seed, stream, n) % np.uint64(hi)).astype(np.int64)


@dataclasses.dataclass
class Scene:
    name: str
    data: np.ndarray        # [N,6] f32
    weak_label: np.ndarray  # [N,2] i64
    seg: np.ndarray         # [N] i32, segment number = rank of the segment's first point
    adj: np.ndarray         # [E0,2] i64
    unmap: np.ndarray       # [V] i64
    gt: np.ndarray          # [V,2] i64

    @property
    def num_points(self) -> int:
        return int(self.data.shape[0])

    @property
    def num_segments(self) -> int:
        return int(self.seg.max()) + 1 if self.seg.size else 0

    def seg_lists(self):
        """The .seg.json payload: list i is non-empty iff i is the first point of a segment."""
        n = self.num_points
        order = np.argsort(self.seg, kind="stable")
        counts = np.bincount(self.seg, minlength=self.num_segments)
        out = [[] for _ in range(n)]
        pos = 0
        for c in counts:
            members = order[pos:pos + c]
            out[int(members[0])] = members.tolist()
            pos += c
        return out


def _renumber_by_first_point(lab: np.ndarray) -> np.ndarray:
    """Relabel so that segment numbers ascend with the index of each segment's first point."""
    _, first = np.unique(lab, return_index=True)
    order = np.argsort(first)
    uniq = np.unique(lab)
    remap = np.empty(int(uniq.max()) + 1, dtype=np.int32)
    remap[uniq[order]] = np.arange(order.size, dtype=np.int32)
    return remap[lab]


def make_scene(num_points: int = 150000, num_segments: int = 1500, seed: int = 0, *,
               name: Optional[str] = None, knn_edges: int = 6, dup_frac: float = 0.0,
               raw_vertices: Optional[int] = None, min_seg: int = 6, island_radius: float = 0.0,
               seg_profile: str = "voronoi") -> Scene:
    """Build one synthetic scene.

    seg_profile = "scannet" switches to `make_scannet_scene` (surfaces, heavy-tailed segment sizes, V != N).

    dup_frac > 0 overwrites that fraction of points with copies of other points (exact
    duplicates exercise the FPS / kNN tie quirks).
    raw_vertices = V != N adds a non-identity `unmap` (every resampled point is hit at
    least once when V >= N; extra raw vertices map to pseudo-random points).
    island_radius > 0 moves every point within that radius of point 0 by +20 m in x BEFORE the
    segmentation / adjacency are built and strips the weak labels there: an unlabeled, disconnected
    component that ends up as cluster 0, i.e. the one situation in which group_unlabeled_clusters has to
    fall back to FPS-1024 nearest-labelled-cluster merging (model.py:479-494).
    """
    from scipy.spatial import cKDTree

    if seg_profile == "scannet":
        return make_scannet_scene(num_points, num_segments, seed, name=name, knn_edges=knn_edges)
    if seg_profile != "voronoi":
        raise ValueError(f"unknown seg_profile {seg_profile!r}")
    n, s = int(num_points), int(num_segments)
    u = uniform01(seed, 1, 6 * n).reshape(n, 6)
    data = np.empty((n, 6), dtype=np.float32)
    data[:, 0] = u[:, 0] * np.float32(8.0)
    data[:, 1] = u[:, 1] * np.float32(6.0)
    data[:, 2] = u[:, 2] * np.float32(3.0)
    data[:, 3:] = u[:, 3:] * np.float32(2.0) - np.float32(1.0)
    if dup_frac > 0:
        m = int(n * dup_frac)
        dst = randint(seed, 7, m, n)
        src = randint(seed, 8, m, n)
        data[dst] = data[src]

    island = None
    if island_radius > 0:
        d0 = np.linalg.norm(data[:, :3].astype(np.float64) - data[0, :3].astype(np.float64), axis=1)
        island = d0 < island_radius
        data[island, 0] += np.float32(20.0)
        # island points first: every island segment then has a smaller root id than any other cluster, so the
        # (merged, isolated, unlabeled) island ends up as cluster 0, whose all-1000 distance row picks itself
        order = np.argsort(~island, kind="stable")
        data = data[order]
        island = island[order]

    xyz = data[:, :3].astype(np.float64)
    tree = cKDTree(xyz)

    # Voronoi over-segmentation around S seed points drawn from the cloud
    attempt = 0
    while True:
        picks = np.unique(randint(seed, 100 + attempt, 4 * s, n))
        perm = np.argsort(splitmix64(seed, 200 + attempt, picks.size), kind="stable")
        seeds = picks[perm][:s]
        _, lab = cKDTree(xyz[seeds]).query(xyz, k=1, workers=-1)
        counts = np.bincount(lab, minlength=s)
        if seeds.size == s and counts.min() >= min_seg:
            break
        attempt += 1
        if attempt > 50:
            raise RuntimeError("could not draw segment seeds with the requested minimum size")
    seg = _renumber_by_first_point(lab.astype(np.int64))

    # mesh-like adjacency: symmetrised k-NN edges, sorted pairs, unique rows
    _, nb = tree.query(xyz, k=knn_edges + 1, workers=-1)
    src = np.repeat(np.arange(n, dtype=np.int64), knn_edges)
    dst = nb[:, 1:].reshape(-1).astype(np.int64)
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    keep = lo != hi
    key = np.unique(lo[keep] * np.int64(n) + hi[keep])
    adj = np.stack([key // n, key % n], axis=1).astype(np.int64)

    # instances: nearest of K seeds over segment centroids; one labelled segment per instance
    counts = np.bincount(seg, minlength=s)
    cent = np.stack([np.bincount(seg, weights=xyz[:, d], minlength=s) for d in range(3)], axis=1) / counts[:, None]
    k_ins = max(2, s // 25)
    ins_seed = np.argsort(splitmix64(seed, 300, s), kind="stable")[:k_ins]
    _, seg_ins = cKDTree(cent[ins_seed]).query(cent, k=1)
    ins_sem = randint(seed, 400, k_ins, 40)
    weak = np.full((n, 2), -1, dtype=np.int64)
    for k in range(k_ins):
        segs = np.nonzero(seg_ins == k)[0]
        if segs.size == 0:
            continue
        big = segs[np.argmax(counts[segs])]  # first largest
        mask = seg == big
        weak[mask, 0] = ins_sem[k]
        weak[mask, 1] = k

    if island is not None:
        weak[island] = -1
    gt_pts = np.stack([ins_sem[seg_ins[seg]] + 1, seg_ins[seg] + 1], axis=1).astype(np.int64)
    if raw_vertices is None or raw_vertices == n:
        unmap = np.arange(n, dtype=np.int64)
    else:
        v = int(raw_vertices)
        unmap = randint(seed, 500, v, n)
        if v >= n:
            slots = np.argsort(splitmix64(seed, 501, v), kind="stable")[:n]
            unmap[slots] = np.arange(n, dtype=np.int64)
    gt = gt_pts[unmap]
    # a few unannotated vertices (sem == 0) so that evaluate()'s valid mask is exercised
    hole = randint(seed, 600, max(1, unmap.size // 50), unmap.size)
    gt = gt.copy()
    gt[hole] = 0

    return Scene(name=name or f"scene{seed:04d}_00", data=data, weak_label=weak, seg=seg.astype(np.int32),
                 adj=adj, unmap=unmap, gt=gt)


def make_scannet_scene(num_points: int = 150000, num_segments: int = 1500, seed: int = 0, *, name: Optional[str] = None,
                       knn_edges: int = 6) -> Scene:
    """A scene shaped like a ScanNet scan rather than like the uniform box of `make_scene` (SURVEY.md 8d calibration):

    * points lie on SURFACES: a floor, four walls and a few hundred small randomly oriented rectangles ("furniture"), with
      ~1 cm of noise off the plane;
    * the over-segmentation is heavy-tailed like ScanNet's normal-based segmentor: the floor is 1-2 segments (10k-30k
      points), every wall 1-2 (3k-8k), the rest of the S segments are Voronoi cells inside the small surfaces (median ~60
      points);
    * K ~ S / 47 instances, ~1.3 labelled segments per instance (manual_label.zip statistics);
    * V != N: even seeds are "sub-sampled" scans (V = 1.2 N raw vertices, several map to one point), odd seeds are "tiled"
      scans (V = 0.85 N: 15 % of the points are exact duplicates of other points, dataset/scannet/util.py:669-681).

    NumPy + SciPy + the counter-based splitmix64 stream only: the GPU box regenerates identical bytes."""
    from scipy.spatial import cKDTree

    n, s = int(num_points), int(num_segments)
    if s < 24 or n < 40 * s // 4:
        raise ValueError("make_scannet_scene needs >= 24 segments and >= 10 points per segment on average")
    tiled = bool(seed & 1)
    n_base = int(n * 0.85) if tiled else n                      # distinct points; a tiled scan repeats some of them
    # ---- structures: (fraction of the points, number of segments) ----
    n_small = max(8, s // 6)                                     # small surfaces ("furniture"), ~6 segments each
    frac = np.concatenate([[0.20], np.full(4, 0.0625), np.full(n_small, 0.55 / n_small)])
    jitter = 0.5 + uniform01(seed, 2, frac.size).astype(np.float64)
    frac = frac * jitter
    frac /= frac.sum()
    struct = np.searchsorted(np.cumsum(frac), uniform01(seed, 3, n_base).astype(np.float64), side="right").clip(0, frac.size - 1)
    u = uniform01(seed, 1, 6 * n_base).reshape(n_base, 6).astype(np.float64)
    xyz = np.empty((n_base, 3), np.float64)
    noise = (u[:, 2] - 0.5) * 0.02
    # floor
    m = struct == 0
    xyz[m] = np.stack([u[m, 0] * 8.0, u[m, 1] * 6.0, noise[m]], 1)
    # walls: x = 0, x = 8, y = 0, y = 6
    for w in range(4):
        m = struct == 1 + w
        a, b = u[m, 0], u[m, 1] * 3.0
        if w < 2:
            xyz[m] = np.stack([np.full(a.shape, 8.0 * w) + noise[m], a * 6.0, b], 1)
        else:
            xyz[m] = np.stack([a * 8.0, np.full(a.shape, 6.0 * (w - 2)) + noise[m], b], 1)
    # small rectangles: centre, two in-plane axes (random orientation), extents 0.15-1.0 m
    pr = uniform01(seed, 4, 9 * n_small).reshape(n_small, 9).astype(np.float64)
    cen = pr[:, :3] * np.array([7.0, 5.0, 2.0]) + np.array([0.5, 0.5, 0.3])
    e1 = pr[:, 3:6] - 0.5
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True) + 1e-9
    tmp = np.cross(e1, np.array([0.0, 0.0, 1.0]))
    tmp[np.linalg.norm(tmp, axis=1) < 1e-6] = np.array([1.0, 0.0, 0.0])
    e2 = tmp / np.linalg.norm(tmp, axis=1, keepdims=True)
    e3 = np.cross(e1, e2)
    ext = 0.15 + 0.85 * pr[:, 6:8]
    m = struct >= 5
    k = struct[m] - 5
    xyz[m] = (cen[k] + e1[k] * ((u[m, 0] - 0.5) * ext[k, 0])[:, None] + e2[k] * ((u[m, 1] - 0.5) * ext[k, 1])[:, None] + e3[k] * noise[m][:, None])
    rgb = u[:, 3:] * 2.0 - 1.0

    # ---- over-segmentation: segments per structure, Voronoi cells inside each structure ----
    counts_struct = np.bincount(struct, minlength=frac.size)
    segs_struct = np.zeros(frac.size, np.int64)
    segs_struct[0] = 1 + (seed >> 1) % 2                         # floor: 1-2 segments
    segs_struct[1:5] = 1 + (splitmix64(seed, 5, 4) % np.uint64(2)).astype(np.int64)
    rest = s - int(segs_struct[:5].sum())
    w_small = counts_struct[5:].astype(np.float64)
    alloc = np.maximum(1, np.floor(w_small / max(w_small.sum(), 1.0) * rest)).astype(np.int64)
    order_small = np.argsort(-w_small, kind="stable")
    i = 0
    while alloc.sum() < rest:
        alloc[order_small[i % n_small]] += 1; i += 1
    while alloc.sum() > rest:
        j = order_small[i % n_small]
        if alloc[j] > 1:
            alloc[j] -= 1
        i += 1
    segs_struct[5:] = alloc
    lab = np.full(n_base, -1, np.int64)
    base = 0
    for st in range(frac.size):
        idx = np.nonzero(struct == st)[0]
        ks = int(min(segs_struct[st], max(idx.size, 1)))
        if idx.size == 0:
            continue
        pick = np.argsort(splitmix64(seed, 1000 + st, idx.size), kind="stable")[:ks]
        _, cell = cKDTree(xyz[idx[pick]]).query(xyz[idx], k=1)
        lab[idx] = base + cell
        base += ks
    # empty structures / merged cells may leave the count short: split the largest segments until there are s of them
    lab = np.unique(lab, return_inverse=True)[1]
    while int(lab.max()) + 1 < s:
        cnt = np.bincount(lab)
        big = int(np.argmax(cnt))
        idx = np.nonzero(lab == big)[0]
        half = xyz[idx, int(np.argmax(np.ptp(xyz[idx], axis=0)))]
        lab[idx[half > np.median(half)]] = int(lab.max()) + 1
    data_base = np.concatenate([xyz, rgb], 1).astype(np.float32)

    # ---- tiled scans: the sampled cloud repeats vertices (exact duplicates, same segment) ----
    if tiled:
        src = np.concatenate([np.arange(n_base, dtype=np.int64), randint(seed, 7, n - n_base, n_base)])
        perm = np.argsort(splitmix64(seed, 8, n), kind="stable")
        src = src[perm]
    else:
        src = np.arange(n, dtype=np.int64)
    data = data_base[src]
    seg = _renumber_by_first_point(lab[src])
    xyz_n = data[:, :3].astype(np.float64)

    # mesh-like adjacency: symmetrised k-NN edges, sorted pairs, unique rows (duplicates are at distance 0 of each other)
    _, nb = cKDTree(xyz_n).query(xyz_n, k=knn_edges + 1, workers=-1)
    a = np.repeat(np.arange(n, dtype=np.int64), knn_edges)
    b = nb[:, 1:].reshape(-1).astype(np.int64)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    keep = lo != hi
    key = np.unique(lo[keep] * np.int64(n) + hi[keep])
    adj = np.stack([key // n, key % n], axis=1).astype(np.int64)

    # ---- instances: K ~ S / 47 seeds over segment centroids; ~1.3 labelled segments per instance ----
    s_real = int(seg.max()) + 1
    counts = np.bincount(seg, minlength=s_real)
    cent = np.stack([np.bincount(seg, weights=xyz_n[:, d], minlength=s_real) for d in range(3)], axis=1) / counts[:, None]
    k_ins = max(2, s_real // 47)
    ins_seed = np.argsort(splitmix64(seed, 300, s_real), kind="stable")[:k_ins]
    _, seg_ins = cKDTree(cent[ins_seed]).query(cent, k=1)
    ins_sem = randint(seed, 400, k_ins, 40)
    weak = np.full((n, 2), -1, dtype=np.int64)
    extra = uniform01(seed, 401, k_ins)
    for k in range(k_ins):
        segs = np.nonzero(seg_ins == k)[0]
        if segs.size == 0:
            continue
        by_size = segs[np.argsort(-counts[segs], kind="stable")]
        chosen = by_size[:2] if (extra[k] < 0.3 and by_size.size > 1) else by_size[:1]
        for sg in chosen:
            mask = seg == sg
            weak[mask, 0] = ins_sem[k]
            weak[mask, 1] = k
    gt_pts = np.stack([ins_sem[seg_ins[seg]] + 1, seg_ins[seg] + 1], axis=1).astype(np.int64)
    # ---- raw vertices: sub-sampled scans have more vertices than points, tiled ones fewer ----
    v = int(n * 1.2) if not tiled else n_base
    if tiled:
        # vertex j is base point j; it maps to the LAST sampled copy of itself (generate_pointcloud_pth, util.py:687-689)
        unmap = np.zeros(v, np.int64)
        unmap[src] = np.arange(n, dtype=np.int64)
    else:
        unmap = randint(seed, 500, v, n)
        slots = np.argsort(splitmix64(seed, 501, v), kind="stable")[:n]
        unmap[slots] = np.arange(n, dtype=np.int64)
    gt = gt_pts[unmap].copy()
    hole = randint(seed, 600, max(1, v // 50), v)
    gt[hole] = 0
    return Scene(name=name or f"scan{seed:04d}_00", data=data, weak_label=weak, seg=seg.astype(np.int32), adj=adj, unmap=unmap, gt=gt)


def write_reference_tree(root: str, scenes, label_style: str = "manual") -> None:
    """Write scenes in the reference's on-disk layout under `root`.

    root/dataset/scannet/scannetv2_train.txt                          (seggroup/model.py:669-672)
    root/dataset/scannet/data/resampled/<s>/<s>.{pcl,info,unmap}.pth  (seggroup/data.py:31-33, model.py:698)
    root/dataset/scannet/label/seg/<style>/resampled/<s>/<s>.label.pth (data.py:32)
    root/dataset/scannet/label/real/resampled/<s>/<s>.seg.json        (model.py:699)
    root/dataset/scannet/label/real/raw/<s>/<s>.label.pth             (model.py:610-611)
    root/dataset/scannet/adj/mesh/resampled/<s>/<s>.adj.pth           (model.py:696)
    """
    import torch

    base = os.path.join(root, "dataset", "scannet")
    os.makedirs(base, exist_ok=True)
    with open(os.path.join(base, "scannetv2_train.txt"), "w") as f:
        for sc in scenes:
            f.write(sc.name + "\n")
    for i, sc in enumerate(scenes):
        d = os.path.join(base, "data", "resampled", sc.name)
        os.makedirs(d, exist_ok=True)
        torch.save(torch.from_numpy(sc.data), os.path.join(d, sc.name + ".pcl.pth"))
        torch.save(torch.tensor([i], dtype=torch.long), os.path.join(d, sc.name + ".info.pth"))
        torch.save(torch.from_numpy(sc.unmap), os.path.join(d, sc.name + ".unmap.pth"))
        d = os.path.join(base, "label", "seg", label_style, "resampled", sc.name)
        os.makedirs(d, exist_ok=True)
        torch.save(torch.from_numpy(sc.weak_label), os.path.join(d, sc.name + ".label.pth"))
        d = os.path.join(base, "label", "real", "resampled", sc.name)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, sc.name + ".seg.json"), "w") as f:
            json.dump(sc.seg_lists(), f)
        d = os.path.join(base, "label", "real", "raw", sc.name)
        os.makedirs(d, exist_ok=True)
        torch.save(torch.from_numpy(sc.gt), os.path.join(d, sc.name + ".label.pth"))
        d = os.path.join(base, "adj", "mesh", "resampled", sc.name)
        os.makedirs(d, exist_ok=True)
        torch.save(torch.from_numpy(sc.adj), os.path.join(d, sc.name + ".adj.pth"))


# ---------------------------------------------------------------------------------------------------------
# Raw-scan side: what dataset/scannet/util.py reads from a ScanNet scan -- the mesh
# (`_vh_clean_2.ply`: vertices xyz + rgb, triangle faces) and the over-segmentation (`segs.json: segIndices`).
# ---------------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class RawScan:
    name: str
    xyz: np.ndarray          # [V,3] float32 (PLY property type)
    rgb: np.ndarray          # [V,3] uint8
    faces: np.ndarray        # [F,3] int32 vertex ids (a few degenerate: repeated ids)
    seg_indices: np.ndarray  # [V]   int32 raw segment ids (non-contiguous, like ScanNet's)
    perm: np.ndarray         # [V]   int64 the permutation `generate_pointcloud_pth` draws with torch.randperm


def make_raw_scan(grid_w: int, grid_h: int, seed: int, *, name: Optional[str] = None, dup_frac: float = 0.02,
                  degenerate_faces: int = 16, cell: int = 7) -> RawScan:
    """A jittered, wavy W x H vertex lattice triangulated with two faces per cell, plus `dup_frac` duplicated
    vertices (same coordinates, referenced by extra faces: coincident vertices give exact distance ties) and a few
    degenerate faces (`get_adj_from_mesh` drops their zero-length edges, util.py:783).  Segments are cell x cell
    vertex blocks with ids 7*b + 3 (np.unique has to compact them)."""
    w, h = int(grid_w), int(grid_h)
    v0 = w * h
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    u = uniform01(seed, 40, 3 * v0).reshape(v0, 3).astype(np.float64)
    x = gx.reshape(-1) * 0.04 + (u[:, 0] - 0.5) * 0.02
    y = gy.reshape(-1) * 0.04 + (u[:, 1] - 0.5) * 0.02
    z = 0.3 * np.sin(x * 1.7) * np.cos(y * 1.3) + (u[:, 2] - 0.5) * 0.01
    xyz = np.stack([x, y, z], 1).astype(np.float32)
    rgb = (uniform01(seed, 41, 3 * v0) * 256).astype(np.int64).clip(0, 255).astype(np.uint8).reshape(v0, 3)
    vid = np.arange(v0, dtype=np.int64).reshape(h, w)
    a, b, c, d = vid[:-1, :-1].reshape(-1), vid[:-1, 1:].reshape(-1), vid[1:, :-1].reshape(-1), vid[1:, 1:].reshape(-1)
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)], 0)
    seg = ((gy.reshape(-1).astype(np.int64) // cell) * ((w + cell - 1) // cell) + gx.reshape(-1).astype(np.int64) // cell) * 7 + 3
    ndup = int(v0 * dup_frac)
    if ndup:
        src = randint(seed, 42, ndup, v0)
        xyz = np.concatenate([xyz, xyz[src]], 0)
        rgb = np.concatenate([rgb, rgb[src]], 0)
        seg = np.concatenate([seg, seg[src]], 0)
        # every duplicate is wired into the mesh by one face with two of its original's neighbours
        nb1 = np.minimum(src + 1, v0 - 1)
        nb2 = np.minimum(src + w, v0 - 1)
        faces = np.concatenate([faces, np.stack([v0 + np.arange(ndup), nb1, nb2], 1)], 0)
    v = xyz.shape[0]
    if degenerate_faces:
        p = randint(seed, 43, 2 * degenerate_faces, v).reshape(-1, 2)
        faces = np.concatenate([faces, np.stack([p[:, 0], p[:, 0], p[:, 1]], 1), np.stack([p[:1, 0], p[:1, 0], p[:1, 0]], 1)], 0)
    forder = np.argsort(splitmix64(seed, 44, faces.shape[0]), kind="stable")          # faces in no particular order
    perm = np.argsort(splitmix64(seed, 45, v), kind="stable").astype(np.int64)
    return RawScan(name or f"scan{seed:04d}_00", xyz, rgb, faces[forder].astype(np.int32), seg.astype(np.int32), perm)


# ------------------------------------------------------------------------------------------------
# synthetic ANNOTATIONS of a raw scan (what ScanNet ships next to the mesh), for the label producers of seggroup_amd/labels.py
# ------------------------------------------------------------------------------------------------
_CATEGORIES = (("wall", 1), ("floor", 2), ("cabinet", 3), ("bed", 4), ("chair", 5), ("sofa", 6), ("table", 7), ("office chair", 5),
               ("desk", 14), ("trash can", 39), ("object", 40))


def make_annotations(scan: "RawScan", seed: int, blocks_per_row: int):
    """-> dict(aggregation=<.aggregation.json content>, tsv=<scannetv2-labels.combined.tsv text>, manual=<click file content>).
    `blocks_per_row` = ceil(grid_w / cell) of make_raw_scan.  Objects are 3 x 2 blocks of the scan's segment grid; every fifth object
    loses its middle column (two disconnected parts of two segments), every fifth + 2 keeps one corner segment apart from the
    rest (a small second part), ~12 % of the segments stay unlabeled, one segment is listed by two objects (the later one wins,
    util.py:123-125).  The manual file clicks one or two segments per object (raw segment ids; strings are allowed too)."""
    seg = np.asarray(scan.seg_indices, dtype=np.int64)
    uniq = np.unique(seg)
    blocks = (uniq - 3) // 7                                  # make_raw_scan: id = 7 * block + 3
    by, bx = blocks // blocks_per_row, blocks % blocks_per_row
    obj_of = (by // 2) * ((blocks_per_row + 2) // 3) + bx // 3
    drop = uniform01(seed, 60, uniq.shape[0]) < 0.12
    groups, manual = [], {}
    for k, o in enumerate(np.unique(obj_of)):
        mine = (obj_of == o) & ~drop
        if k % 5 == 4:
            mine &= bx % 3 != 1                               # two parts that do not touch
        if k % 5 == 2:
            mine &= ~((bx % 3 == 1) | ((bx % 3 == 2) & (by % 2 == 1)))      # a big part and a single far corner
        members = uniq[mine]
        if members.size == 0:
            continue
        label = _CATEGORIES[k % len(_CATEGORIES)][0]
        groups.append({"id": k, "objectId": k, "segments": [int(s) for s in members], "label": label})
        manual[str(k + 1)] = [int(members[0])] if k % 3 else [int(members[0]), str(int(members[-1]))]
    if len(groups) > 2:
        groups[2]["segments"].append(groups[1]["segments"][0])              # claimed twice: the later group overwrites
    tsv = "id\traw_category\tcategory\tcount\tnyu40id\n" + "".join(f"{i}\t{n}\t{n}\t1\t{c}\n" for i, (n, c) in enumerate(_CATEGORIES))
    return dict(aggregation={"sceneId": scan.name, "segGroups": groups}, tsv=tsv, manual=manual)
