"""Packed binary scene cache.

The reference reads, per scene and per forward, three `.pth` files in the DataLoader (seggroup/data.py:34-37)
plus `<s>.adj.pth`, `<s>.seg.json` (a 150k-entry list of lists, ~1.5 MB of JSON), the GT `.label.pth`
and -- 14 times -- `<s>.unmap.pth` inside `SegModel.forward` (model.py:696-699,714,724,533,612).  That is ~60-80 ms
of parsing per scene, 20x the GPU time of the whole forward here.  A pack is ONE file per (scene, label style):

    magic "SGPACK01" | u32 header length | JSON header {name, N, S, E0, V, arrays: {name: [dtype, shape, offset]}} | raw arrays
    (64-byte aligned, little endian)

holding exactly the staged form `DeviceScene` needs: data f32[N,6], adj i64[E0,2], seg_of_point i32[N], the CSR of
the `.seg.json` lists (seg_points i32[N], seg_off i32[S+1]), unmap i32[V], gt i32[V,2] and the per-segment host arrays
(seg_first, seg_size, seg_ins, seg_sem i32[S]).  `load_pack` memory-maps the file and uploads each array once.
Packs are derived data: they are rebuilt when any source file is newer.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, List, Optional

import numpy as np

MAGIC = b"SGPACK01"
ARRAYS = ("data", "adj", "seg_of_point", "seg_points", "seg_off", "unmap", "gt", "seg_first", "seg_size", "seg_ins", "seg_sem")


def pack_path(root: str, scene_name: str, label_style: str = "manual") -> str:
    return os.path.join(root, "dataset", "scannet", "cache", label_style, scene_name + ".sgpack")


def source_files(root: str, scene_name: str, label_style: str = "manual") -> List[str]:
    base = os.path.join(root, "dataset", "scannet")
    return [os.path.join(base, "data", "resampled", scene_name, scene_name + ".pcl.pth"),
            os.path.join(base, "data", "resampled", scene_name, scene_name + ".unmap.pth"),
            os.path.join(base, "label", "seg", label_style, "resampled", scene_name, scene_name + ".label.pth"),
            os.path.join(base, "label", "real", "resampled", scene_name, scene_name + ".seg.json"),
            os.path.join(base, "label", "real", "raw", scene_name, scene_name + ".label.pth"),
            os.path.join(base, "adj", "mesh", "resampled", scene_name, scene_name + ".adj.pth")]


def stage_arrays(data, weak_label, seg, adj, unmap, gt) -> Dict[str, np.ndarray]:
    """The staged arrays of one scene from the reference's tensors (same derivation as DeviceScene.__init__)."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    weak_label = np.asarray(weak_label, dtype=np.int64)
    seg = np.ascontiguousarray(seg, dtype=np.int32)
    n = data.shape[0]
    if data.shape[1] != 6 or weak_label.shape != (n, 2) or seg.shape != (n,):
        raise ValueError("stage_arrays: inconsistent input shapes")
    s = int(seg.max()) + 1
    # CSR of the over-segmentation + first point / size of every segment: one native counting pass (sg_stage_segments)
    from . import hip
    order = np.empty(n, dtype=np.int32)
    off = np.empty(s + 1, dtype=np.int32)
    first = np.empty(s, dtype=np.int32)
    counts = np.empty(s, dtype=np.int32)
    try:
        hip.check(hip.lib().sg_stage_segments(seg.ctypes.data, n, s, order.ctypes.data, off.ctypes.data, first.ctypes.data, counts.ctypes.data))
    except hip.SgError as e:
        raise ValueError(str(e)) from None
    return dict(data=data, adj=np.ascontiguousarray(np.asarray(adj, dtype=np.int64).reshape(-1, 2)), seg_of_point=seg,
                seg_points=order, seg_off=off, unmap=np.ascontiguousarray(unmap, dtype=np.int32),
                gt=np.ascontiguousarray(gt, dtype=np.int32), seg_first=first, seg_size=counts,
                seg_ins=np.ascontiguousarray(weak_label[first, 1], dtype=np.int32),
                seg_sem=np.ascontiguousarray(weak_label[first, 0], dtype=np.int32))


def write_pack(path: str, name: str, arrays: Dict[str, np.ndarray], adj_int32: bool = True) -> None:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    meta, blobs, off = {}, [], 0
    for k in ARRAYS:
        a = np.ascontiguousarray(arrays[k])
        if adj_int32 and k == "adj" and a.dtype == np.int64 and (a.size == 0 or (0 <= int(a.min()) and int(a.max()) < 2 ** 31)):
            # point indices fit int32: half of a pack's bytes were this array as int64 (8.6 of 17 MB at 150k points / 536k edges).  Readers
            # hand out int64 again: read_pack widens on the host, load_pack / the native loader on the device
            a = a.astype(np.int32)
        meta[k] = [a.dtype.str, list(a.shape), off]
        blobs.append(a)
        off += (a.nbytes + 63) // 64 * 64
    hdr = json.dumps({"name": name, "N": int(arrays["data"].shape[0]), "S": int(arrays["seg_first"].shape[0]),
                      "E0": int(arrays["adj"].shape[0]), "V": int(arrays["unmap"].shape[0]), "arrays": meta}).encode()
    pad = (64 - (len(MAGIC) + 4 + len(hdr)) % 64) % 64
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(hdr) + pad))
        f.write(hdr + b" " * pad)
        for a in blobs:
            f.write(memoryview(a).cast("B"))
            f.write(b"\0" * ((64 - a.nbytes % 64) % 64))
    os.replace(tmp, path)                       # atomic: concurrent ranks may race to build the same pack


def read_pack(path: str) -> Dict[str, object]:
    """-> {'name', 'N', 'S', 'E0', 'V', arrays...} with the arrays as (private, copy-on-write) memory maps."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a SegGroup scene pack")
        (hlen,) = struct.unpack("<I", f.read(4))
        hdr = json.loads(f.read(hlen).decode())
    base = 12 + hlen
    mm = np.memmap(path, dtype=np.uint8, mode="c")      # copy-on-write: torch wants writable buffers; nothing writes
    out = {k: hdr[k] for k in ("name", "N", "S", "E0", "V")}
    for k, (dt, shape, off) in hdr["arrays"].items():
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[k] = mm[base + off: base + off + n].view(np.dtype(dt)).reshape(shape)
    if out["adj"].dtype != np.int64:
        out["adj"] = out["adj"].astype(np.int64)            # stored as int32 (write_pack)
    return out


def pack_scene(root: str, scene_name: str, label_style: str = "manual", force: bool = False) -> str:
    """Build (or reuse) the pack of one scene of the reference's on-disk tree; returns its path."""
    from .pth import load_tensor
    from .scene import seg_from_file

    path = pack_path(root, scene_name, label_style)
    src = source_files(root, scene_name, label_style)
    if not force and os.path.exists(path) and os.path.getmtime(path) >= max(os.path.getmtime(p) for p in src):
        return path

    def ld(p):                                   # torch-free reader first (no torch import in pool workers)
        try:
            return load_tensor(p)
        except Exception:
            import torch
            return torch.load(p, map_location="cpu").numpy()
    data, unmap, weak = ld(src[0]), ld(src[1]), ld(src[2])
    seg = seg_from_file(src[3], data.shape[0])
    gt, adj = ld(src[4]), ld(src[5])
    write_pack(path, scene_name, stage_arrays(data, weak, seg, adj, unmap, gt))
    return path


def is_current(root: str, scene_name: str, label_style: str = "manual") -> bool:
    """True if the scene's pack exists and no source file is newer."""
    path = pack_path(root, scene_name, label_style)
    try:
        return os.path.getmtime(path) >= max(os.path.getmtime(p) for p in source_files(root, scene_name, label_style))
    except OSError:
        return False


def _pack_job(job):
    return pack_scene(*job)


def build_missing(root: str, scene_names, label_style: str = "manual", workers: int = 8) -> int:
    """Build the packs that are missing or stale with `workers` THREADS: with the native seg.json parser, the native
    segment staging and the torch-free .pth reader a pack is ~10 ms of work that releases the GIL (file reads, ctypes calls,
    NumPy copies, the pack write), so threads scale and nothing pays a fresh interpreter's ~2 s torch import (round 1 used
    spawned processes: 48 scenes/s cold).  Returns the number of packs built."""
    todo = [n for n in scene_names if not is_current(root, n, label_style)]
    if not todo:
        return 0
    # Round 6: the native builder first (csrc/packbuild.cpp: plain threads, no interpreter in the loop -- the Python path below is ~2,000
    # interpreter-level calls per scene under the GIL, ~260 scenes/s whatever the thread count); scenes it refuses (a container it does not
    # know) are built in Python as before.  SG_PACK_BUILD=python skips it.
    n_todo = len(todo)
    if os.environ.get("SG_PACK_BUILD", "native") != "python":
        todo = _build_native(root, todo, label_style, workers)
        if not todo:
            return n_todo
    if len(todo) < 4 or workers <= 1:
        for n in todo:
            pack_scene(root, n, label_style)
        return n_todo
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(workers, len(todo))) as pool:
        list(pool.map(_pack_job, [(root, n, label_style) for n in todo]))
    return n_todo


def _build_native(root: str, todo, label_style: str, workers: int):
    """sg_pack_build_many over `todo`; returns the scenes that still need the Python builder."""
    import ctypes as C
    from . import hip
    lib = hip.lib()
    n = len(todo)
    srcs, outs = [], []
    for name in todo:
        srcs += source_files(root, name, label_style)
        out = pack_path(root, name, label_style)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        outs.append(out)
    a_src = (C.c_char_p * (6 * n))(*[p.encode() for p in srcs])
    a_names = (C.c_char_p * n)(*[t.encode() for t in todo])
    a_out = (C.c_char_p * n)(*[p.encode() for p in outs])
    status = (C.c_int32 * n)()
    rc = lib.sg_pack_build_many(a_src, a_names, a_out, n, max(1, int(workers)), status)
    if rc < 0:
        return list(todo)
    return [name for name, st in zip(todo, status) if st != 0]


_tls = __import__("threading").local()


def _pinned(nbytes: int):
    """A per-thread pinned staging buffer (grown on demand): loader threads reuse theirs for every scene."""
    import torch
    buf = getattr(_tls, "pinned", None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 24), dtype=torch.uint8, pin_memory=True)
        _tls.pinned = buf
    return buf


def pack_dims(path: str) -> Dict[str, int]:
    """{'N', 'S', 'E0', 'V'} from a pack's header (a few hundred bytes read)."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a SegGroup scene pack")
        (hlen,) = struct.unpack("<I", f.read(4))
        hdr = json.loads(f.read(hlen).decode())
    return {k: int(hdr[k]) for k in ("N", "S", "E0", "V")}


def load_pack(path: str, device="cuda"):
    """Pack -> DeviceScene with ONE host-to-device copy: the file is read into a pinned buffer, uploaded as one blob, and
    the device arrays are typed views into it (every array starts on a 64-byte boundary of the file)."""
    import torch
    from .scene import DeviceScene
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a SegGroup scene pack")
        (hlen,) = struct.unpack("<I", f.read(4))
        hdr = json.loads(f.read(hlen).decode())
        base = 12 + hlen
        size = os.fstat(f.fileno()).st_size - base
        pin = _pinned(size)
        view = pin.numpy()[:size]
        got = f.readinto(memoryview(view))
        if got != size:
            raise ValueError(f"{path}: truncated pack")
    dev = torch.device(device)
    blob = torch.empty(size, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        stream = getattr(_tls, "stream", None)
        if stream is None or stream.device != dev:
            stream = _tls.stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            blob.copy_(pin[:size], non_blocking=True)
        # the four small per-segment host arrays are copied out of the staging buffer before it is reused
        host = {}
        for k in ("seg_first", "seg_size", "seg_ins", "seg_sem"):
            dt, shape, off = hdr["arrays"][k]
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            host[k] = view[off:off + n].view(np.dtype(dt)).reshape(shape).copy()
        from .scene import seg_of_vertex

        def hv(k):
            dt, shape, off = hdr["arrays"][k]
            return view[off:off + int(np.prod(shape)) * np.dtype(dt).itemsize].view(np.dtype(dt)).reshape(shape)
        sov = seg_of_vertex(hv("seg_of_point"), hv("unmap"))          # host-side look-up table of the compact label transfer
        stream.synchronize()                                  # the pinned buffer is free again, the blob is complete
    tdt = {"<f4": torch.float32, "<i4": torch.int32, "<i8": torch.int64}
    arrays = {}
    for k in ARRAYS:
        dt, shape, off = hdr["arrays"][k]
        if k in host:
            arrays[k] = host[k]
        else:
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            arrays[k] = blob[off:off + n].view(tdt[dt]).view(*shape)
    if arrays["adj"].dtype != torch.int64:
        arrays["adj"] = arrays["adj"].to(torch.int64)       # stored as int32 (write_pack): widened on the device
    arrays["seg_of_vertex"] = sov
    return DeviceScene.from_staged(arrays, name=hdr["name"], device=dev)


class LoadedScene:
    """A scene resident in a loader slot: what the engine needs of a DeviceScene (`c_struct`, the dimensions, the name, the host-side
    seg_of_vertex), with the memory owned by the native loader until `release()`."""

    def __init__(self, loader: "PackLoader", c_scene, slot: int, name: str):
        import ctypes as C
        self._loader, self._c, self.slot, self.name = loader, c_scene, slot, name
        self.device = loader.device
        self.N, self.S, self.E0, self.V = int(c_scene.N), int(c_scene.S), int(c_scene.E0), int(c_scene.V)
        self.h_seg_of_vertex = np.ctypeslib.as_array(C.cast(c_scene.h_seg_of_vertex, C.POINTER(C.c_int32)), shape=(self.V,))
        self.h_seg_size = np.ctypeslib.as_array(C.cast(c_scene.h_seg_size, C.POINTER(C.c_int32)), shape=(self.S,))

    @property
    def c_struct(self):
        return self._c

    @property
    def released(self) -> bool:
        return self.slot is None

    def release(self) -> None:
        """Hand the slot back.  `h_seg_of_vertex` / `h_seg_size` are views into the slot's host storage, which the next load overwrites:
        they are withdrawn here, and a SceneResult that still wants to expand from them raises (model.SceneResult.labels)."""
        if self.slot is not None:
            self._loader.release(self.slot)
            self.slot = None
            self.h_seg_of_vertex = self.h_seg_size = None


class PackLoader:
    """`sg_loader_*` (csrc/loader.cpp): native threads read scene packs into pinned buffers and upload them into pre-allocated device
    slots.  submit(path) -> ticket at once; wait(ticket) -> LoadedScene; the scene's slot is free again after LoadedScene.release()."""

    def __init__(self, threads: int, slots: int, slot_bytes: int, device=None, max_edges: int = 0, copy_limit: int = 0):
        import torch
        from . import hip
        hip.require_device()
        self.lib = hip.lib()
        self.device = torch.device(device if device is not None else "cuda")
        with torch.cuda.device(self.device):
            self.handle = self.lib.sg_loader_create_sized(int(threads), int(slots), int(slot_bytes), int(max_edges))
        if not self.handle:
            raise hip.SgError(hip.SG_ENOMEM, self.lib.sg_last_error().decode())
        if copy_limit > 0:
            hip.check(self.lib.sg_loader_set_copy_limit(self.handle, int(copy_limit)))

    def submit(self, path: str) -> int:
        from . import hip
        t = self.lib.sg_loader_submit(self.handle, path.encode())
        hip.check(t)
        return t

    def wait(self, ticket: int) -> LoadedScene:
        import ctypes as C
        from . import hip
        sc = hip.Scene()
        slot = C.c_int(-1)
        name = C.create_string_buffer(256)
        hip.check(self.lib.sg_loader_wait(self.handle, ticket, C.byref(sc), C.byref(slot), name, 256))
        return LoadedScene(self, sc, slot.value, name.value.decode())

    def release(self, slot: int) -> None:
        from . import hip
        hip.check(self.lib.sg_loader_release(self.handle, slot))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.sg_loader_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
