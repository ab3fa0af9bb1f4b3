"""Raw ScanNet scan -> the hot path's input files.

Host-side mirror of the parts of the reference's offline pre-processing that produce what `SegModel.forward` reads
(`seggroup/dataset/scannet/util.py`, driven by `prepare_data.py:36-71` and `prepare_weak_label.py`): same function
names, same files, same bytes -- the compute runs on the GPU through the C ABI (`sg_prep_sample_points`,
`sg_nearest_point`, `sg_mesh_adjacency`, `sg_segment_lists`), there is no CPU path.

    generate_pointcloud_pth          util.py:633-693   data/resampled/<s>/<s>.{pcl,info,map,unmap}.pth
    generate_seg_labels_and_ds_set   util.py:174-220   label/real/raw/<s>/<s>.seg.txt, label/real/resampled/<s>/<s>.seg.json
    generate_mesh_adjcency_pth       util.py:795-811   adj/mesh/{raw,resampled}/<s>/<s>.adj.pth
    get_unmapper / get_adj_from_mesh util.py:538-550 / 771-792

Paths are relative to `root` (the reference uses the working directory).  `plydata` may be a `plyfile.PlyData`
(not installed here) or the `PlyMesh` of `read_ply` below: only `['vertex'][name]`, `['face']['vertex_indices']`
and `.count` are used.  Ground-truth / weak-label generation (util.py:129-170, 268-427, 697-768) needs ScanNet's
annotation files and stays out of scope.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional

import numpy as np

from . import hip


# ---- minimal PLY (binary little endian; ScanNet's `_vh_clean_2.ply`) ---------------------------------------------
_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "<i2", "ushort": "<u2", "int": "<i4", "uint": "<u4", "float": "<f4", "double": "<f8",
              "int8": "i1", "uint8": "u1", "int16": "<i2", "uint16": "<u2", "int32": "<i4", "uint32": "<u4", "float32": "<f4", "float64": "<f8"}


class _Element:
    def __init__(self, count, cols):
        self.count, self._cols = count, cols

    def __getitem__(self, name):
        return self._cols[name]


class PlyMesh:
    """`mesh['vertex']['x']`, `mesh['face']['vertex_indices']` ([F,3] int32), `.count` -- the subset of plyfile used."""

    def __init__(self, elements):
        self._e = elements

    def __getitem__(self, name):
        return self._e[name]


def read_ply(path: str) -> PlyMesh:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                elements[-1][2].append(tok[1:])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: only binary_little_endian PLY is supported (ScanNet's format), got {fmt}")
        out = {}
        for name, count, props in elements:
            if any(p[0] == "list" for p in props):
                if len(props) != 1:
                    raise ValueError(f"{path}: element {name}: a list property next to other properties is not supported")
                _, ctype, itype, pname = props[0]
                rec = np.dtype([("n", _PLY_TYPES[ctype]), ("v", _PLY_TYPES[itype], (3,))])
                raw = np.frombuffer(f.read(rec.itemsize * count), dtype=rec, count=count)
                if count and not (raw["n"] == 3).all():
                    raise ValueError(f"{path}: element {name}: only triangle faces are supported")
                out[name] = _Element(count, {pname: np.ascontiguousarray(raw["v"]).astype(np.int32)})
            else:
                rec = np.dtype([(p[1], _PLY_TYPES[p[0]]) for p in props])
                raw = np.frombuffer(f.read(rec.itemsize * count), dtype=rec, count=count)
                out[name] = _Element(count, {p[1]: np.ascontiguousarray(raw[p[1]]) for p in props})
    return PlyMesh(out)


def write_ply(path: str, xyz, rgb, faces) -> None:
    """ScanNet-shaped binary PLY (vertex: float xyz + uchar rgba; face: uchar count + int ids) -- for tests and tools."""
    xyz, rgb, faces = np.asarray(xyz, np.float32), np.asarray(rgb, np.uint8), np.asarray(faces, np.int32)
    v = np.zeros(xyz.shape[0], dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")])
    v["x"], v["y"], v["z"], v["red"], v["green"], v["blue"], v["alpha"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], rgb[:, 0], rgb[:, 1], rgb[:, 2], 255
    fc = np.zeros(faces.shape[0], dtype=[("n", "u1"), ("v", "<i4", (3,))])
    fc["n"], fc["v"] = 3, faces
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\nelement face %d\n"
           "property list uchar int vertex_indices\nend_header\n" % (xyz.shape[0], faces.shape[0]))
    with open(path, "wb") as f:
        f.write(hdr.encode())
        f.write(v.tobytes())
        f.write(fc.tobytes())


def mesh_arrays(plydata):
    """-> xyz f32 [V,3], rgb u8 [V,3], faces i32 [F,3] from a PlyMesh / plyfile.PlyData."""
    vtx, face = plydata["vertex"], plydata["face"]
    xyz = np.stack([np.asarray(vtx[k], dtype=np.float32) for k in ("x", "y", "z")], 1)
    rgb = np.stack([np.asarray(vtx[k], dtype=np.uint8) for k in ("red", "green", "blue")], 1)
    vi = face["vertex_indices"]
    faces = np.asarray(vi, dtype=np.int32) if isinstance(vi, np.ndarray) and vi.ndim == 2 else \
        np.stack([np.asarray(x, dtype=np.int32) for x in vi]) if len(vi) else np.zeros((0, 3), np.int32)
    if faces.ndim != 2 or faces.shape[1] != 3:
        raise ValueError("only triangle meshes are supported")
    return np.ascontiguousarray(xyz), np.ascontiguousarray(rgb), np.ascontiguousarray(faces)


def load_seg_labels(label_file):                               # util.py:95-100
    with open(label_file, "r") as f:
        return json.load(f)["segIndices"]


# ---- device plumbing --------------------------------------------------------------------------------------------
def _dev(device=None):
    import torch
    hip.require_device()
    return torch.device(device if device is not None else "cuda")


def _ws(nbytes, dev):
    import torch
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def _t(a, dtype, dev):
    import torch
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)


def get_unmapper(x, y, device=None):
    """util.py:538-550: for every row of x [U,3] the index of its nearest row of y [N,3] (LongTensor on the device)."""
    import torch
    dev = _dev(device)
    lib = hip.lib()
    dx, dy = _t(x, torch.float32, dev), _t(y, torch.float32, dev)
    out = torch.empty(dx.shape[0], dtype=torch.int64, device=dev)
    ws = _ws(lib.sg_nearest_point_ws_bytes(dy.shape[0]), dev)
    with torch.cuda.device(dev):
        hip.check(lib.sg_nearest_point(dx.data_ptr(), dx.shape[0], dy.data_ptr(), dy.shape[1], dy.shape[0], out.data_ptr(), ws.data_ptr(),
                                       ws.numel(), None))
        torch.cuda.synchronize()
    return out


def sample_points(xyz, rgb, mapper, device=None):
    """The compute of generate_pointcloud_pth: -> (pointcloud_sampled [Np,6] f32, unmapper [V] i64, #unsampled vertices)."""
    import torch
    dev = _dev(device)
    lib = hip.lib()
    d_xyz, d_rgb, d_map = _t(xyz, torch.float32, dev), _t(rgb, torch.uint8, dev), _t(mapper, torch.int64, dev)
    v, n = d_xyz.shape[0], d_map.shape[0]
    pcl = torch.empty((n, 6), dtype=torch.float32, device=dev)
    unmap = torch.empty(v, dtype=torch.int64, device=dev)
    ws = _ws(lib.sg_prep_sample_ws_bytes(v, n), dev)
    miss = C.c_int(0)
    with torch.cuda.device(dev):
        hip.check(lib.sg_prep_sample_points(d_xyz.data_ptr(), d_rgb.data_ptr(), v, d_map.data_ptr(), n, pcl.data_ptr(), unmap.data_ptr(),
                                            C.byref(miss), ws.data_ptr(), ws.numel(), None))
        torch.cuda.synchronize()
    return pcl, unmap, miss.value


def mesh_adjacency(faces, unmapper=None, num_vertices: Optional[int] = None, device=None):
    """The compute of get_adj_from_mesh: -> (adj [E,2] i64, adj_resampled [E',2] i64 or None), device tensors."""
    import torch
    dev = _dev(device)
    lib = hip.lib()
    d_f = _t(faces, torch.int32, dev)
    f = d_f.shape[0]
    d_un = _t(unmapper, torch.int64, dev) if unmapper is not None else None
    v = int(num_vertices if num_vertices is not None else (d_un.shape[0] if d_un is not None else int(d_f.max().item()) + 1 if f else 1))
    raw = torch.empty((max(3 * f, 1), 2), dtype=torch.int64, device=dev)
    res = torch.empty((max(3 * f, 1), 2), dtype=torch.int64, device=dev) if d_un is not None else None
    ws = _ws(lib.sg_mesh_adjacency_ws_bytes(f), dev)
    n_raw, n_res = C.c_int(0), C.c_int(0)
    with torch.cuda.device(dev):
        hip.check(lib.sg_mesh_adjacency(d_f.data_ptr(), f, d_un.data_ptr() if d_un is not None else None, v, raw.data_ptr(), C.byref(n_raw),
                                        res.data_ptr() if res is not None else None, C.byref(n_res) if res is not None else None,
                                        ws.data_ptr(), ws.numel(), None))
        torch.cuda.synchronize()
    return raw[:n_raw.value], (res[:n_res.value] if res is not None else None)


def segment_lists(seg_indices, mapper, device=None):
    """The compute of generate_seg_labels_and_ds_set: -> (raw_label [V] i32, seg_points [Np] i32, seg_off [G+1] i32) device
    tensors: groups in ascending compacted-id order, members ascending."""
    import torch
    dev = _dev(device)
    lib = hip.lib()
    d_seg, d_map = _t(seg_indices, torch.int32, dev), _t(mapper, torch.int64, dev)
    v, n = d_seg.shape[0], d_map.shape[0]
    raw = torch.empty(v, dtype=torch.int32, device=dev)
    pts = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.empty(min(v, n) + 1, dtype=torch.int32, device=dev)
    ws = _ws(lib.sg_segment_lists_ws_bytes(v, n), dev)
    counts = (C.c_int * 2)()
    with torch.cuda.device(dev):
        hip.check(lib.sg_segment_lists(d_seg.data_ptr(), v, d_map.data_ptr(), n, raw.data_ptr(), pts.data_ptr(), off.data_ptr(), counts,
                                       ws.data_ptr(), ws.numel(), None))
        torch.cuda.synchronize()
    return raw, pts, off[:counts[1] + 1]


def get_adj_from_mesh(plydata, unmapper=None, device=None):
    """util.py:771-792 -> (adj, adj_resampled) LongTensors (CPU, like the reference's)."""
    _, _, faces = mesh_arrays(plydata)
    raw, res = mesh_adjacency(faces, unmapper, num_vertices=plydata["vertex"].count, device=device)
    return raw.cpu(), (res.cpu() if res is not None else None)


def get_adj_from_pointcloud(pointcloud, k=10, device=None):
    """util.py:814-834 (optional in the reference: nothing calls it): the kNN graph of the cloud as a per-row sorted, unique
    [*, 2] LongTensor (CPU, like the reference's).  Equal scores rank by ascending index (torch.topk leaves that open)."""
    import torch
    dev = _dev(device)
    lib = hip.lib()
    pts = _t(pointcloud, torch.float32, dev)
    n, stride = int(pts.shape[0]), int(pts.shape[1])
    out = torch.empty((n * int(k), 2), dtype=torch.int64, device=dev)
    ws = _ws(lib.sg_pointcloud_adjacency_ws_bytes(n, int(k)), dev)
    cnt = C.c_int(0)
    with torch.cuda.device(dev):
        hip.check(lib.sg_pointcloud_adjacency(pts.data_ptr(), stride, n, int(k), out.data_ptr(), C.byref(cnt), ws.data_ptr(), ws.numel(), None))
        torch.cuda.synchronize()
    return out[:cnt.value].cpu()


# ---- the reference's file-producing functions -------------------------------------------------------------------------
def _scene_name(scene_path: str) -> str:
    return os.path.split(scene_path[:-1] if scene_path.endswith("/") else scene_path)[-1]


def make_mapper(num_vertices: int, num_points: int, perm=None):
    """util.py:664-676.  `perm` replaces the `torch.randperm(V)` draw (pass one for reproducible output)."""
    import torch
    rep, rem = num_points // num_vertices, num_points % num_vertices
    if rem > 0:
        p = torch.as_tensor(np.asarray(perm), dtype=torch.long) if perm is not None else torch.randperm(num_vertices)
        index_remainder = p[:rem]
    else:
        index_remainder = torch.LongTensor([])
    if rep != 0:
        return torch.cat([torch.arange(num_vertices).repeat(rep), index_remainder], dim=0)
    return index_remainder


def generate_pointcloud_pth(scene_path, item, num_points, plydata=None, root: str = ".", perm=None, device=None):
    """util.py:633-693: `.pcl.pth` f32 [num_points,6], `.info.pth`, `.map.pth`, `.unmap.pth` under data/resampled/<scene>/."""
    import torch
    scene_name = _scene_name(scene_path)
    if plydata is None:
        plydata = read_ply(os.path.join(scene_path, scene_name + "_vh_clean_2.ply"))
    xyz, rgb, _ = mesh_arrays(plydata)
    mapper = make_mapper(xyz.shape[0], num_points, perm)
    pcl, unmap, _ = sample_points(xyz, rgb, mapper, device=device)
    out = os.path.join(root, "data", "resampled", scene_name)
    os.makedirs(out, exist_ok=True)
    torch.save(pcl.cpu(), os.path.join(out, scene_name + ".pcl.pth"))
    torch.save(torch.LongTensor([item]), os.path.join(out, scene_name + ".info.pth"))
    torch.save(mapper, os.path.join(out, scene_name + ".map.pth"))
    torch.save(unmap.cpu(), os.path.join(out, scene_name + ".unmap.pth"))


def generate_seg_labels_and_ds_set(scene_path, root: str = ".", device=None):
    """util.py:174-220: label/real/raw/<s>/<s>.seg.txt and label/real/resampled/<s>/<s>.seg.json."""
    import torch
    scene_name = _scene_name(scene_path)
    seg = np.asarray(load_seg_labels(os.path.join(scene_path, scene_name + "_vh_clean_2.0.010000.segs.json")), dtype=np.int64)
    if seg.size and seg.min() < 0:
        raise ValueError("segIndices must be non-negative")
    mapper = torch.load(os.path.join(root, "data", "resampled", scene_name, scene_name + ".map.pth"))
    raw, pts, off = segment_lists(seg.astype(np.int32), mapper, device=device)
    lib = hip.lib()
    d1 = os.path.join(root, "label", "real", "raw", scene_name)
    os.makedirs(d1, exist_ok=True)
    h_raw = np.ascontiguousarray(raw.cpu().numpy())
    hip.check(lib.sg_write_label_txt(os.path.join(d1, scene_name + ".seg.txt").encode(), h_raw.ctypes.data, h_raw.shape[0]))
    d2 = os.path.join(root, "label", "real", "resampled", scene_name)
    os.makedirs(d2, exist_ok=True)
    h_pts, h_off = np.ascontiguousarray(pts.cpu().numpy()), np.ascontiguousarray(off.cpu().numpy())
    hip.check(lib.sg_write_seg_json(os.path.join(d2, scene_name + ".seg.json").encode(), h_pts.ctypes.data, h_off.ctypes.data,
                                    h_off.shape[0] - 1, h_pts.shape[0]))


def generate_mesh_adjcency_pth(scene_name, plydata=None, root: str = ".", scene_path: Optional[str] = None, device=None):
    """util.py:795-811: adj/mesh/raw/<s>/<s>.adj.pth and adj/mesh/resampled/<s>/<s>.adj.pth (skipped when both exist)."""
    import torch
    p1 = os.path.join(root, "adj", "mesh", "raw", scene_name, scene_name + ".adj.pth")
    p2 = os.path.join(root, "adj", "mesh", "resampled", scene_name, scene_name + ".adj.pth")
    if os.path.exists(p1) and os.path.exists(p2):
        return
    if plydata is None:
        if scene_path is None:
            raise ValueError("generate_mesh_adjcency_pth: pass plydata or scene_path")
        plydata = read_ply(os.path.join(scene_path, scene_name + "_vh_clean_2.ply"))
    unmapper = torch.load(os.path.join(root, "data", "resampled", scene_name, scene_name + ".unmap.pth"))
    adj, adj_resampled = get_adj_from_mesh(plydata, unmapper, device=device)
    for p, t in ((p1, adj), (p2, adj_resampled)):
        os.makedirs(os.path.dirname(p), exist_ok=True)
        torch.save(t, p)


# the annotation-derived label producers of the same reference file (util.py:76-170, 224-427, 697-768) live in labels.py
from .labels import (generate_real_label_pth, generate_real_labels, generate_seg_adjacency_matrix, generate_weak_label_pth,  # noqa: E402,F401
                     generate_weak_labels, group_adjacency_segs, load_aggregation, load_labels, read_label_mapper)


def prepare_scene(scene_path, item, num_points: int = 150000, root: str = ".", perm=None, device=None, label_style: Optional[str] = None,
                  manual_label_path: Optional[str] = None):
    """What prepare_data.py:36-71 + prepare_weak_label.py:60-90 do for one scan: point cloud, mapper / unmapper, segment lists,
    mesh adjacency and -- with `label_style` and ScanNet's annotation files next to the mesh -- the ground-truth and weak-label
    files, i.e. every input of SegModel.forward."""
    scene_name = _scene_name(scene_path)
    ply = read_ply(os.path.join(scene_path, scene_name + "_vh_clean_2.ply"))
    generate_pointcloud_pth(scene_path, item, num_points, ply, root=root, perm=perm, device=device)
    generate_seg_labels_and_ds_set(scene_path, root=root, device=device)
    generate_mesh_adjcency_pth(scene_name, ply, root=root, device=device)
    if label_style is not None:
        generate_real_labels(scene_path, root=root)
        generate_real_label_pth(scene_path, root=root)
        generate_weak_labels(scene_path, ply, label_style=label_style, manual_label_path=manual_label_path, root=root)
        generate_weak_label_pth(scene_name, label_style, root=root)
