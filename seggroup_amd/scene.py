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


def seg_from_file(path: str, num_points: int) -> np.ndarray:
    """`.seg.json` file -> segment number per point through the native parser (`sg_parse_seg_json`: same checks as
    `seg_from_lists`, without building 150k Python lists first)."""
    seg = np.empty(num_points, dtype=np.int32)
    hip.check(hip.lib().sg_parse_seg_json(path.encode(), int(num_points), seg.ctypes.data))
    return seg


def seg_of_vertex(seg_of_point: np.ndarray, unmap: np.ndarray) -> np.ndarray:
    """The over-segment of every raw vertex, seg_of_point[unmap[v]] (-1 where unmap[v] is not a point): what a label vector is looked
    up through (model.py:525-605); the engine's compact label transfer does that look-up on the host."""
    seg_of_point = np.asarray(seg_of_point)
    p_ = np.asarray(unmap).astype(np.int64, copy=False)
    ok = (p_ >= 0) & (p_ < seg_of_point.shape[0])
    out = np.full(p_.shape[0], -1, dtype=np.int32)
    out[ok] = seg_of_point[p_[ok]]
    return out


class DeviceScene:
    """One scene resident on a HIP device (torch tensors) + the segment-level host arrays."""

    def __init__(self, data, weak_label, seg, adj, unmap, gt, device="cuda", name: str = "scene"):
        from .cache import stage_arrays
        self._from_staged(stage_arrays(data, weak_label, seg, adj, unmap, gt), name, device)

    @classmethod
    def from_staged(cls, arrays, name: str = "scene", device="cuda") -> "DeviceScene":
        """From the staged arrays of `cache.stage_arrays` / a scene pack (no recomputation, one upload per array)."""
        self = cls.__new__(cls)
        self._from_staged(arrays, name, device)
        return self

    def _from_staged(self, a, name, device):
        import torch

        self.name = name
        self.N, self.S = int(a["data"].shape[0]), int(a["seg_first"].shape[0])
        self.E0, self.V = int(a["adj"].shape[0]), int(a["unmap"].shape[0])
        # host, segment level (contiguous copies: the C ABI keeps raw pointers to them)
        self.h_seg_first = np.ascontiguousarray(a["seg_first"], dtype=np.int32)
        self.h_seg_size = np.ascontiguousarray(a["seg_size"], dtype=np.int32)
        self.h_seg_ins = np.ascontiguousarray(a["seg_ins"], dtype=np.int32)
        self.h_seg_sem = np.ascontiguousarray(a["seg_sem"], dtype=np.int32)
        sov = a.get("seg_of_vertex")
        if sov is None and not isinstance(a["seg_of_point"], torch.Tensor) and not isinstance(a["unmap"], torch.Tensor):
            sov = seg_of_vertex(a["seg_of_point"], a["unmap"])
        self.h_seg_of_vertex = None if sov is None else np.ascontiguousarray(sov, dtype=np.int32)
        dev = torch.device(device)
        self.device = dev
        # device tensors pass through (views into a scene pack's blob); host arrays are uploaded
        host = lambda x: (lambda a: a if a.flags.writeable else a.copy())(np.ascontiguousarray(x))     # read-only views (np.load, mmap) are copied
        up = lambda x: x if isinstance(x, torch.Tensor) else torch.from_numpy(host(x)).to(dev, non_blocking=False)
        self.d_data = up(a["data"])
        self.d_adj = up(a["adj"])
        self.d_seg_of_point = up(a["seg_of_point"])
        self.d_seg_points = up(a["seg_points"])
        self.d_seg_off = up(a["seg_off"])
        self.d_unmap = up(a["unmap"])
        self.d_gt = up(a["gt"])
        self._c = hip.Scene(N=self.N, S=self.S, E0=self.E0, V=self.V,
                            d_data=self.d_data.data_ptr(), d_adj=self.d_adj.data_ptr(),
                            d_seg_of_point=self.d_seg_of_point.data_ptr(), d_seg_points=self.d_seg_points.data_ptr(),
                            d_seg_off=self.d_seg_off.data_ptr(), d_unmap=self.d_unmap.data_ptr(), d_gt=self.d_gt.data_ptr(),
                            h_seg_first=self.h_seg_first.ctypes.data, h_seg_size=self.h_seg_size.ctypes.data,
                            h_seg_ins=self.h_seg_ins.ctypes.data, h_seg_sem=self.h_seg_sem.ctypes.data,
                            h_seg_of_vertex=None if self.h_seg_of_vertex is None else self.h_seg_of_vertex.ctypes.data)

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
        from .pth import load_tensor

        base = os.path.join(root, "dataset", "scannet")

        def ld(*p):
            try:
                return load_tensor(os.path.join(base, *p))
            except Exception:
                import torch
                return torch.load(os.path.join(base, *p), map_location="cpu").numpy()
        if data is None:
            data = ld("data", "resampled", scene_name, scene_name + ".pcl.pth")
        if weak_label is None:
            weak_label = ld("label", "seg", label_style, "resampled", scene_name, scene_name + ".label.pth")
        data = data.detach().cpu().numpy() if hasattr(data, "detach") else np.asarray(data)
        weak_label = weak_label.detach().cpu().numpy() if hasattr(weak_label, "detach") else np.asarray(weak_label)
        adj = ld("adj", "mesh", "resampled", scene_name, scene_name + ".adj.pth")
        unmap = ld("data", "resampled", scene_name, scene_name + ".unmap.pth")
        gt = ld("label", "real", "raw", scene_name, scene_name + ".label.pth")
        seg = seg_from_file(os.path.join(base, "label", "real", "resampled", scene_name, scene_name + ".seg.json"), data.shape[0])
        return cls(data, weak_label, seg, adj, unmap, gt, device=device, name=scene_name)
