"""Scene staging: the reference's per-scene inputs -> device-resident arrays of the C ABI.

What the reference reads for one scene: `data[N,6]`, `weak_label[N,2]` from the
DataLoader (seggroup/data.py:28-38), and by scene name inside forward: `<s>.adj.pth`, `<s>.unmap.pth`,
`<s>.seg.json` (seggroup/model.py:696-699,714,724) and the GT `<s>.label.pth` (model.py:610-612).
`DeviceScene` is the staged form of exactly that: the JSON member lists become a CSR
(`seg_points`/`seg_off`, ascending point index per segment) and the per-segment weak labels are the
weak labels of each segment's first point (DisjointSet init, model.py:712-721).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Optional

import numpy as np

from . import hip


def seg_from_lists(lists, num_points: int) -> np.ndarray:
    """`.seg.json` payload (list i non-empty iff i is a segment's first point) -> segment number per point."""
    seg = np.full(num_points, -1, dtype=np.int32)
    s = 0
    for i, members in enumerate(lists):
        if members:
            if members[0] != i:
                raise ValueError(f"seg.json list {i} does not start at its own index (got {members[0]})")
            seg[np.asarray(members, dtype=np.int64)] = s
            s += 1
    if (seg < 0).any():
        raise ValueError("seg.json does not cover every point (the reference raises KeyError in update_adj here)")
    return seg


class DeviceScene:
    """One scene resident on a HIP device (torch tensors) + the segment-level host arrays."""

    def __init__(self, data, weak_label, seg, adj, unmap, gt, device="cuda", name: str = "scene"):
        import torch

        data = np.ascontiguousarray(data, dtype=np.float32)
        weak_label = np.asarray(weak_label, dtype=np.int64)
        seg = np.ascontiguousarray(seg, dtype=np.int32)
        adj = np.ascontiguousarray(np.asarray(adj, dtype=np.int64).reshape(-1, 2))
        unmap = np.ascontiguousarray(unmap, dtype=np.int32)
        gt = np.ascontiguousarray(gt, dtype=np.int32)
        n = data.shape[0]
        if data.shape[1] != 6 or weak_label.shape != (n, 2) or seg.shape != (n,):
            raise ValueError("DeviceScene: inconsistent input shapes")
        s = int(seg.max()) + 1
        order = np.argsort(seg, kind="stable").astype(np.int32)     # points ascending inside each segment
        counts = np.bincount(seg, minlength=s).astype(np.int32)
        off = np.zeros(s + 1, dtype=np.int32)
        np.cumsum(counts, out=off[1:])
        first = order[off[:-1]].astype(np.int32)
        if s > 1 and not (np.diff(first) > 0).all():
            raise ValueError("segment numbers must ascend with each segment's first point (see synthetic._renumber_by_first_point)")
        self.name = name
        self.N, self.S, self.E0, self.V = n, s, int(adj.shape[0]), int(unmap.shape[0])
        # host, segment level
        self.h_seg_first = first
        self.h_seg_size = counts
        self.h_seg_ins = np.ascontiguousarray(weak_label[first, 1], dtype=np.int32)
        self.h_seg_sem = np.ascontiguousarray(weak_label[first, 0], dtype=np.int32)
        # device
        dev = torch.device(device)
        self.device = dev
        up = lambda a: torch.from_numpy(a).to(dev)
        self.d_data = up(data)
        self.d_adj = up(adj)
        self.d_seg_of_point = up(seg)
        self.d_seg_points = up(order)
        self.d_seg_off = up(off)
        self.d_unmap = up(unmap)
        self.d_gt = up(gt)
        self._c = hip.Scene(N=self.N, S=self.S, E0=self.E0, V=self.V,
                            d_data=self.d_data.data_ptr(), d_adj=self.d_adj.data_ptr(),
                            d_seg_of_point=self.d_seg_of_point.data_ptr(), d_seg_points=self.d_seg_points.data_ptr(),
                            d_seg_off=self.d_seg_off.data_ptr(), d_unmap=self.d_unmap.data_ptr(), d_gt=self.d_gt.data_ptr(),
                            h_seg_first=self.h_seg_first.ctypes.data, h_seg_size=self.h_seg_size.ctypes.data,
                            h_seg_ins=self.h_seg_ins.ctypes.data, h_seg_sem=self.h_seg_sem.ctypes.data)

    @property
    def c_struct(self) -> hip.Scene:
        return self._c

    @classmethod
    def from_synthetic(cls, sc, device="cuda") -> "DeviceScene":
        return cls(sc.data, sc.weak_label, sc.seg, sc.adj, sc.unmap, sc.gt, device=device, name=sc.name)

    @classmethod
    def from_reference_tree(cls, scene_name: str, data=None, weak_label=None, root: str = ".", label_style: str = "manual",
                            device="cuda") -> "DeviceScene":
        """Read one scene from the reference's on-disk layout under `root` (CWD in the reference)."""
        import torch

        base = os.path.join(root, "dataset", "scannet")
        ld = lambda *p: torch.load(os.path.join(base, *p), map_location="cpu")
        if data is None:
            data = ld("data", "resampled", scene_name, scene_name + ".pcl.pth")
        if weak_label is None:
            weak_label = ld("label", "seg", label_style, "resampled", scene_name, scene_name + ".label.pth")
        data = data.detach().cpu().numpy() if hasattr(data, "detach") else np.asarray(data)
        weak_label = weak_label.detach().cpu().numpy() if hasattr(weak_label, "detach") else np.asarray(weak_label)
        adj = ld("adj", "mesh", "resampled", scene_name, scene_name + ".adj.pth").numpy()
        unmap = ld("data", "resampled", scene_name, scene_name + ".unmap.pth").numpy()
        gt = ld("label", "real", "raw", scene_name, scene_name + ".label.pth").numpy()
        with open(os.path.join(base, "label", "real", "resampled", scene_name, scene_name + ".seg.json")) as f:
            seg = seg_from_lists(json.load(f), data.shape[0])
        return cls(data, weak_label, seg, adj, unmap, gt, device=device, name=scene_name)
