"""SegGroup model, MI355X-native: the drop-in for the reference's `seggroup/model.py`.

Same surface: `SegModel(exp_name, cuda, visualize, sem_infer, ins_infer)`, attribute
`.epoch`, `forward(data[1,N,6], weak_label[1,N,2], info[1,1]) -> (IoU_sem[1,2,40], IoU_ins[1,2,40],
acc[4])`, the reference's checkpoint key layout, CWD-relative dataset paths, and the
`results/<exp>/<scene>/<tag>/layer_*.{seg,ins,sem}.txt` + `final.{ins,sem}.txt` outputs (plus a
`.npy` twin of every file).  Underneath, one C-ABI call (`sg_pipeline_forward`) runs the whole
forward on the GPU; there is no CPU fallback.

The network modules below only hold parameters (so `load_state_dict` / DDP work unchanged); their
arithmetic lives in seggroup_amd/csrc/*.hip.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .scene import DeviceScene
from . import weights as _weights
from . import functional as _F
# module-level functions of the reference's model.py, same names: the operator seam
from .functional import (DisjointSet, aggregate_cluster_feature, build_distance_matrix, build_similarity_matrix,  # noqa: F401
                         calculate_distance, calculate_similarity, combine_centralized_pointcloud, evaluate,
                         export_instance_label, export_segment_label, export_semantic_label, farthest_point_sampling,
                         get_cluster_pointcloud, get_knn, group_nearby_clusters, knn, l2_norm, update_adj)

SEM_VALID_CLASS_IDS = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])
INS_VALID_CLASS_IDS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39])


# ---- parameter containers with the reference's module / key structure (model.py:65-166) ----------
class MLP1(nn.Module):
    def __init__(self):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(64)
        self.conv1 = nn.Sequential(nn.Conv2d(6, 64, kernel_size=1, bias=False), self.bn1, nn.LeakyReLU(negative_slope=0.2))

    def forward(self, x):                                  # [S,6,64] -> [S,128]  (model.py:73-80)
        return _F.mlp1_forward(x, self.conv1[0].weight, self.bn1.weight, self.bn1.bias)


class MLP2(nn.Module):
    def __init__(self):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(64)
        self.conv1 = nn.Sequential(nn.Conv2d(18, 64, kernel_size=1, bias=False), self.bn1, nn.LeakyReLU(negative_slope=0.2))

    def forward(self, x, idx):                             # [1,9,N], [1,N,20] -> [1,64,N]  (model.py:114-118)
        return _F.edgeconv_forward(x, idx, self.conv1[0].weight, self.bn1.weight, self.bn1.bias)


class MLP3(nn.Module):
    def __init__(self):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(64)
        self.conv1 = nn.Sequential(nn.Conv2d(18, 64, kernel_size=1, bias=False), self.bn1, nn.LeakyReLU(negative_slope=0.2))
        self.bn2 = nn.BatchNorm2d(64)
        self.conv2 = nn.Sequential(nn.Conv2d(64, 64, kernel_size=1, bias=False), self.bn2, nn.LeakyReLU(negative_slope=0.2))

    def forward(self, x, idx):                             # model.py:133-138
        return _F.edgeconv_forward(x, idx, self.conv1[0].weight, self.bn1.weight, self.bn1.bias,
                                   self.conv2[0].weight, self.bn2.weight, self.bn2.bias)


class GCN(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.fc = nn.Linear(dim_in, dim_out, bias=False)

    def forward(self, X, Edge):
        """model.py:146-151.  `Edge` is either the reference's dense similarity matrix [S,S] or -- cheaper -- the
        adjacency list [E,2] it was built from (the similarities are then recomputed sparsely on the device)."""
        if Edge.dim() == 2 and Edge.shape[0] == Edge.shape[1] and Edge.shape[0] == X.shape[0] and Edge.is_floating_point():
            iu = torch.nonzero(torch.triu(Edge, diagonal=1))
            return _F.gcn_forward(X, iu, self.fc.weight)
        return _F.gcn_forward(X, Edge, self.fc.weight)


class Classifier(nn.Module):
    """Train-only head (model.py:154-166); kept so that reference checkpoints load with strict=True."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.linear1 = nn.Linear(dim_in, 128, bias=False)
        self.bn1 = nn.BatchNorm1d(128)
        self.dp1 = nn.Dropout(p=0.5)
        self.linear2 = nn.Linear(128, dim_out)


class SceneResult:
    """Outputs of one forward: 14 (ins mode) or 6 (sem mode) int32 label vectors + metric tensors."""

    def __init__(self, labels: Optional[np.ndarray], n_vectors: int, res: hip.Result, tables: Optional[np.ndarray] = None,
                 seg_of_vertex: Optional[np.ndarray] = None, owner=None):
        self._labels = labels                     # [14, V] int32 (pinned) -- or None: compact transfer, expanded on first use
        self.tables = tables                      # [14, S] int32 (compact transfer only)
        self._sov = seg_of_vertex
        # a scene in a native loader slot (cache.LoadedScene) lends its seg_of_vertex: the view dies with `release()`.  The result
        # keeps the scene and refuses to expand from a released slot -- `detach()` takes a private copy first.
        self._owner = owner if hasattr(owner, "released") else None
        self.n_vectors = n_vectors
        self.iou_sem = np.ctypeslib.as_array(res.iou_sem).reshape(1, 2, 40).copy()
        self.iou_ins = np.ctypeslib.as_array(res.iou_ins).reshape(1, 2, 40).copy()
        self.acc = np.ctypeslib.as_array(res.acc).copy()
        self.trace = list(res.trace)
        self.stalled = bool(res.stalled)
        self.used_fallback = bool(res.used_fallback)
        self.feat5 = self.ins5 = self.sem5 = None        # train mode only: Feat_5 [C5,256] + weak labels of the final clusters

    @property
    def labels(self) -> np.ndarray:
        if self._labels is None:                  # compact transfer: labels[t][v] = tables[t][seg_of_vertex[v]] (-1 outside), on the host
            if self._owner is not None and self._owner.released:
                raise RuntimeError("SceneResult.labels: the scene's loader slot was released before the label vectors were expanded "
                                   "(read .labels or call .detach() before LoadedScene.release())")
            nv, S = self.tables.shape
            out = np.empty((nv, self._sov.shape[0]), dtype=np.int32)
            hip.check(hip.lib().sg_expand_labels(self.tables.ctypes.data, nv, S, self._sov.ctypes.data, self._sov.shape[0], out.ctypes.data))
            self._labels = out
        return self._labels

    @labels.setter
    def labels(self, v):
        self._labels = v

    def detach(self) -> "SceneResult":
        """Make the result independent of the loader slot its scene lives in (a private copy of seg_of_vertex, 4 V bytes)."""
        if self._owner is not None and self._labels is None:
            if self._owner.released:
                raise RuntimeError("SceneResult.detach: the scene's loader slot is already released")
            self._sov = np.array(self._sov, copy=True)
        self._owner = None
        return self

    def label_dict(self) -> Dict[str, np.ndarray]:
        return {hip.LABEL_NAMES[i]: self.labels[i] for i in range(self.n_vectors)}


class Pipeline:
    """Owns one `sg_pipeline` (one in-flight scene on one HIP stream) and its pinned output buffer."""

    def __init__(self, w: Dict[str, np.ndarray], max_points: int, max_segments: int, max_edges: int, max_vertices: int,
                 stream: Optional[torch.cuda.Stream] = None, device=None):
        hip.require_device()
        self.lib = hip.lib()
        self.device = torch.device(device if device is not None else "cuda")
        self.stream = stream
        self.caps = (max_points, max_segments, max_edges, max_vertices)
        cw, self._keep = _c_weights(w)
        with torch.cuda.device(self.device):
            sp = stream.cuda_stream if stream is not None else None
            self.handle = self.lib.sg_pipeline_create(max_points, max_segments, max_edges, max_vertices, C.byref(cw), sp)
        if not self.handle:
            raise hip.SgError(hip.SG_EHIP, self.lib.sg_last_error().decode())
        self.labels = torch.empty((hip.NUM_LABEL_VECTORS, max_vertices), dtype=torch.int32, pin_memory=True)

    def fits(self, sc: DeviceScene) -> bool:
        return sc.N <= self.caps[0] and sc.S <= self.caps[1] and sc.E0 <= self.caps[2] and sc.V <= self.caps[3]

    def forward(self, sc: DeviceScene, mode: int = hip.MODE_INS_INFER, debug: Optional[hip.Debug] = None, want_feat5: bool = False) -> SceneResult:
        res = hip.Result()
        # A pinned block of exactly this scene's [14, V] per forward, handed to the result as it is: torch's caching host allocator returns the
        # block of an earlier, dropped result (no hipHostMalloc in steady state), and nothing is copied -- the private copy this replaced was
        # 8.4 MB per 150k-point scene (0.2 ms of a 2.2 ms forward; 0.85 ms at 500k points).  A result stays valid for as long as it is held.
        lab_t = torch.empty((hip.NUM_LABEL_VECTORS, max(sc.V, 1)), dtype=torch.int32, pin_memory=True)
        res.h_labels = lab_t.data_ptr()
        feat5 = ins5 = sem5 = None
        if want_feat5 and mode == hip.MODE_INS_INFER:
            debug = debug if debug is not None else hip.Debug()
            feat5 = np.zeros((sc.S, 256), np.float32); ins5 = np.zeros(sc.S, np.int32); sem5 = np.zeros(sc.S, np.int32)
            debug.h_feat5, debug.h_ins5, debug.h_sem5 = feat5.ctypes.data, ins5.ctypes.data, sem5.ctypes.data
        dbg_ref = C.byref(debug) if debug is not None else None
        if self.device.index is None or torch.cuda.current_device() == self.device.index:      # (the library sets its device itself)
            rc = self.lib.sg_pipeline_forward(self.handle, C.byref(sc.c_struct), mode, C.byref(res), dbg_ref)
        else:
            with torch.cuda.device(self.device):
                rc = self.lib.sg_pipeline_forward(self.handle, C.byref(sc.c_struct), mode, C.byref(res), dbg_ref)
        hip.check(rc)
        nvec = 14 if mode == hip.MODE_INS_INFER else 6
        # the C side packs the vectors at stride V of THIS scene (include/seggroup_hip.h, sg_result.h_labels)
        lab = lab_t.numpy()[:, :sc.V]                              # a view: the array keeps the pinned tensor alive
        out = SceneResult(lab, nvec, res)
        if feat5 is not None:
            n5 = int(debug.n5)
            out.feat5, out.ins5, out.sem5 = feat5[:n5].copy(), ins5[:n5].copy(), sem5[:n5].copy()
        return out

    def set_timing(self, level: int) -> int:
        """0 = no stage-timing events, 1 = only around the kNN / EdgeConv kernels, 2 = every stage (default)."""
        return self.lib.sg_pipeline_set_timing(self.handle, int(level))

    def stage_times(self) -> Dict[str, float]:
        buf = (C.c_float * 32)()
        n = self.lib.sg_pipeline_stage_times(self.handle, buf, 32)
        return {self.lib.sg_pipeline_stage_name(i).decode(): float(buf[i]) for i in range(n)}

    def device_bytes(self) -> int:
        return int(self.lib.sg_pipeline_device_bytes(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sg_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _c_weights(w: Dict[str, np.ndarray]):
    k = {n: np.ascontiguousarray(v, dtype=np.float32) for n, v in w.items()}
    cw = hip.Weights(
        mlp1_w=k["mlp_1.conv1.0.weight"].ctypes.data, mlp1_g=k["mlp_1.bn1.weight"].ctypes.data, mlp1_b=k["mlp_1.bn1.bias"].ctypes.data,
        mlp2_w=k["mlp_2.conv1.0.weight"].ctypes.data, mlp2_g=k["mlp_2.bn1.weight"].ctypes.data, mlp2_b=k["mlp_2.bn1.bias"].ctypes.data,
        gcn2_w=k["gcn_2.fc.weight"].ctypes.data,
        mlp3_w1=k["mlp_3.conv1.0.weight"].ctypes.data, mlp3_g1=k["mlp_3.bn1.weight"].ctypes.data, mlp3_b1=k["mlp_3.bn1.bias"].ctypes.data,
        mlp3_w2=k["mlp_3.conv2.0.weight"].ctypes.data, mlp3_g2=k["mlp_3.bn2.weight"].ctypes.data, mlp3_b2=k["mlp_3.bn2.bias"].ctypes.data,
        gcn3_w=k["gcn_3.fc.weight"].ctypes.data)
    return cw, k


class Ticket:
    """One submitted batch: keeps the ctypes arrays alive until `Engine.wait` and names the label buffer it fills."""

    def __init__(self, tid, scenes, c_scenes, c_res, c_dirs, labels, mode):
        self.id, self.scenes, self.c_scenes, self.c_res, self.c_dirs, self.labels, self.mode = tid, scenes, c_scenes, c_res, c_dirs, labels, mode


class Engine:
    """The scene engine (`sg_engine_*`, csrc/engine.cpp): `groups` native host threads, each advancing up to `per_group`
    scenes in lock-step with ONE launch per kernel and phase for all of them (scene index = grid.y).  `submit()` returns at
    once, `wait()` blocks: a driver that submits batch k+1 before waiting for batch k keeps the GPU busy across batches.

    Label vectors of a waited ticket are VIEWS into one of `ring` pinned buffers used in turn: they stay valid until the
    `ring`-th `submit()` after theirs (ring = 3: two tickets can be queued behind the one being consumed)."""

    def __init__(self, w: Dict[str, np.ndarray], caps, groups: int = 4, per_group: int = 8, device=None, timing: int = 0,
                 label_transfer: str = "full"):
        """label_transfer: "full" = the 14 label vectors of every scene cross PCIe (8.4 MB per 150k-vertex scene); "tables" = only the
        [14,S] tables do, the vectors are looked up on the host (in the writer pool's workers, or on first access of SceneResult.labels)."""
        hip.require_device()
        self.lib = hip.lib()
        self.device = torch.device(device if device is not None else "cuda")
        self.caps = tuple(int(c) for c in caps)
        self.groups, self.per_group = int(groups), int(per_group)
        cw, self._keep = _c_weights(w)
        with torch.cuda.device(self.device):
            self.handle = self.lib.sg_engine_create(*self.caps, C.byref(cw), self.groups, self.per_group)
        if not self.handle:
            raise hip.SgError(hip.SG_EHIP, self.lib.sg_last_error().decode())
        self.lib.sg_engine_set_timing(self.handle, int(timing))
        if label_transfer not in ("full", "tables"):
            raise ValueError("label_transfer: 'full' or 'tables'")
        self.compact = label_transfer == "tables"
        hip.check(self.lib.sg_engine_set_label_transfer(self.handle, 1 if self.compact else 0))
        self.ring = 3
        self._labels = [None] * self.ring
        self._slot_writer = [None] * self.ring            # (writer, ticket id) of the files still being written out of a ring slot
        self._retired = []                                # (writer, ticket id, buffer): label buffers replaced while their ticket was unconsumed
        self._waited = set()                              # ticket ids whose sg_engine_wait has returned (their writer jobs are all queued)
        self._turn = 0
        self._names = [self.lib.sg_pipeline_stage_name(i).decode() for i in range(32) if self.lib.sg_pipeline_stage_name(i)]

    def fits(self, sc: DeviceScene) -> bool:
        return sc.N <= self.caps[0] and sc.S <= self.caps[1] and sc.E0 <= self.caps[2] and sc.V <= self.caps[3]

    def submit(self, scenes: List[DeviceScene], mode: int = hip.MODE_INS_INFER, writer: "Optional[AsyncLabelWriter]" = None,
               out_dirs: Optional[List[str]] = None, formats=("txt", "npy")) -> Ticket:
        n = len(scenes)
        if any(not self.fits(s) for s in scenes):
            raise ValueError("Engine.submit: a scene exceeds the capacities this engine was created with")
        slot = self._turn
        self._turn = (self._turn + 1) % self.ring
        self._release_slot(slot)                          # the writer pool reads the label vectors in place (sg_writer_submit_scene)
        self._drain_retired()
        # compact transfer: no label vectors cross PCIe, so no pinned label ring either (1.6 GB at 3 x 64 scenes of 150k vertices)
        need_labels = not (self.compact and all(s.h_seg_of_vertex is not None for s in scenes))
        if need_labels and (self._labels[slot] is None or self._labels[slot].shape[0] < n):
            # pinning hundreds of MB takes ~0.1 s: every ring slot is (re)sized at once, not one per submit
            for k in range(self.ring):
                if self._labels[k] is None or self._labels[k].shape[0] < n:
                    # A slot other than `slot` may belong to a ticket that is still in flight in the engine: its writer jobs do not exist
                    # yet, so sg_writer_wait_tag would return at once and the pool would later format files out of freed pinned memory
                    # (ADVICE round 3).  Such a buffer is retired, not released: it lives until its ticket was waited for AND its tag drained.
                    pending = self._slot_writer[k]
                    if pending is not None and pending[1] not in self._waited:
                        self._retired.append((pending[0], pending[1], self._labels[k]))
                        self._slot_writer[k] = None
                    else:
                        self._release_slot(k)
                    self._labels[k] = torch.empty((max(n, 1), hip.NUM_LABEL_VECTORS, self.caps[3]), dtype=torch.int32, pin_memory=True)
        buf = self._labels[slot] if need_labels else None
        c_scenes = (hip.Scene * n)(*[s.c_struct for s in scenes])
        c_res = (hip.Result * n)()
        tabs = None
        if self.compact:
            tabs = [np.empty((hip.NUM_LABEL_VECTORS, s.S), dtype=np.int32) for s in scenes]
        for i in range(n):
            c_res[i].h_labels = buf[i].data_ptr() if buf is not None else None
            if tabs is not None:
                c_res[i].h_tables = tabs[i].ctypes.data
        c_dirs, wh, fm = None, None, 0
        if writer is not None and out_dirs is not None:
            for d_ in out_dirs:
                os.makedirs(d_, exist_ok=True)
            c_dirs = (C.c_char_p * n)(*[d_.encode() for d_ in out_dirs])
            wh = writer.handle
            fm = (1 if "txt" in formats else 0) | (2 if "npy" in formats else 0)
        with torch.cuda.device(self.device):
            tid = self.lib.sg_engine_submit(self.handle, c_scenes, n, mode, c_res, wh, c_dirs, fm)
        hip.check(tid)
        if wh is not None:
            self._slot_writer[slot] = (writer, tid)
        t = Ticket(tid, list(scenes), c_scenes, c_res, c_dirs, buf, mode)
        t.tables = tabs
        return t

    def _release_slot(self, slot: int) -> None:
        """Block until the writer pool has written the files whose label vectors still sit in ring slot `slot`."""
        pending = self._slot_writer[slot]
        if pending is not None:
            w, tid = pending
            if getattr(w, "handle", None):
                hip.check(self.lib.sg_writer_wait_tag(w.handle, tid))
            self._slot_writer[slot] = None
            self._waited.discard(tid)

    def _drain_retired(self, everything: bool = False) -> None:
        """Drop retired label buffers whose files are written.  A tag can only be drained once the engine has queued all of the
        ticket's writer jobs, i.e. after `wait()` on it returned (or, with `everything`, after the engine's threads are gone)."""
        keep = []
        for w, tid, buf in self._retired:
            if everything or tid in self._waited:
                if getattr(w, "handle", None):
                    hip.check(self.lib.sg_writer_wait_tag(w.handle, tid))
                self._waited.discard(tid)
            else:
                keep.append((w, tid, buf))
        self._retired = keep

    def wait(self, t: Ticket) -> List[SceneResult]:
        hip.check(self.lib.sg_engine_wait(self.handle, t.id))
        if any(p is not None and p[1] == t.id for p in self._slot_writer) or any(r[1] == t.id for r in self._retired):
            self._waited.add(t.id)
        nvec = 14 if t.mode == hip.MODE_INS_INFER else 6
        lab = t.labels.numpy() if t.labels is not None else None
        nv = hip.NUM_LABEL_VECTORS                  # scene i's vectors are packed at stride V_i inside its slot
        out = []
        for i, s in enumerate(t.scenes):
            if getattr(t, "tables", None) is not None and s.h_seg_of_vertex is not None:
                out.append(SceneResult(None, nvec, t.c_res[i], tables=t.tables[i][:nvec], seg_of_vertex=s.h_seg_of_vertex, owner=s))
            else:
                out.append(SceneResult(lab[i].reshape(-1)[:nv * s.V].reshape(nv, s.V), nvec, t.c_res[i]))
        return out

    def run(self, scenes: List[DeviceScene], mode: int = hip.MODE_INS_INFER, writer: "Optional[AsyncLabelWriter]" = None,
            out_dirs: Optional[List[str]] = None, formats=("txt", "npy")) -> List[SceneResult]:
        """Forward every scene; with `writer` + `out_dirs` the native threads also hand each scene's label vectors to
        the writer pool (files appear asynchronously: call writer.flush())."""
        return self.wait(self.submit(scenes, mode, writer, out_dirs, formats))

    def set_knn_variant(self, variant: int) -> int:
        return self.lib.sg_engine_set_knn_variant(self.handle, int(variant))

    def set_timing(self, level: int) -> int:
        """0 = no stage events, 1 = a handful of HIP events per batched launch sequence (mean_stage_ms); returns the previous level."""
        return self.lib.sg_engine_set_timing(self.handle, int(level))

    def mean_stage_ms(self) -> Dict[str, float]:
        """Device time per stage and SCENE: the duration of every batched launch divided by the scenes it covered."""
        buf = (C.c_double * 32)()
        n = self.lib.sg_engine_stage_times(self.handle, buf, 32, 0)
        return {nm: float(buf[i]) / max(int(n), 1) for i, nm in enumerate(self._names)}

    def reset_stage_stats(self):
        self.lib.sg_engine_stage_times(self.handle, None, 0, 1)
        self.lib.sg_engine_profile(self.handle, None, 1, 0)

    def profile(self, enable: bool = False) -> Dict[str, float]:
        """Where the group threads' wall time went since the last reset (development aid)."""
        out = (C.c_double * 8)()
        self.lib.sg_engine_profile(self.handle, out, 0, 1 if enable else 0)
        n = max(out[0], 1.0)
        return {"super_steps": out[0], "scenes_per_super_step": out[1] / n, "ms_per_super_step": out[2] / n,
                "ms_blocked_in_sync": out[3] / n, "ms_host_work": (out[2] - out[3]) / n, "ms_idle": out[4] / n}

    def device_bytes(self) -> int:
        return int(self.lib.sg_engine_device_bytes(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            # the label buffers die with this object: the writers must be done with them.  Destroy first (the group threads finish
            # their super-step and queue its writer jobs), then drain every tag, retired buffers included
            self.lib.sg_engine_destroy(self.handle)
            self.handle = None
            for k in range(self.ring):
                self._release_slot(k)
            self._drain_retired(everything=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchRunner(Engine):
    """`inflight` scenes in flight on one GPU: the engine with capacities taken from a list of scenes (the driver's and
    the tests' entry point).  inflight >= 16 runs inflight // 8 groups of 8 scenes."""

    def __init__(self, w: Dict[str, np.ndarray], scenes: List[DeviceScene], inflight: int = 4, device=None, min_caps=None,
                 timing: int = 0, per_group: Optional[int] = None, label_transfer: str = "full"):
        dev = torch.device(device if device is not None else scenes[0].device)
        caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
        if min_caps is not None:                    # regrowing: never shrink below the previous capacities
            caps = tuple(max(a, b) for a, b in zip(caps, min_caps))
        groups, per_group = self.shape(inflight, per_group)
        super().__init__(w, caps, groups=groups, per_group=per_group, device=dev, timing=timing, label_transfer=label_transfer)

    @staticmethod
    def shape(inflight: int, per_group: Optional[int] = None):
        """(groups, scenes per group) for `inflight` scenes in flight: groups of 8 from 16 scenes on.  The ONE statement of that rule: the driver,
        which creates its engine before the first batch has arrived, asks here too (it restated the formula by hand)."""
        if per_group is None:
            per_group = min(8, max(1, inflight // 2))
        return max(1, inflight // per_group), per_group


class AsyncLabelWriter:
    """Pool of native threads writing label files (`sg_writer_*`): submit() copies the vectors and returns."""

    def __init__(self, threads: int = 4, max_queue: int = 256):
        self.lib = hip.lib()
        self.handle = self.lib.sg_writer_create(threads, max_queue)
        if not self.handle:
            raise hip.SgError(hip.SG_EINVAL, self.lib.sg_last_error().decode())

    def submit(self, output_root: str, result: "SceneResult", formats=("txt", "npy")) -> None:
        os.makedirs(output_root, exist_ok=True)
        fm = (1 if "txt" in formats else 0) | (2 if "npy" in formats else 0)
        for i in range(result.n_vectors):
            vec = np.ascontiguousarray(result.labels[i])
            hip.check(self.lib.sg_writer_submit(self.handle, os.path.join(output_root, hip.LABEL_NAMES[i]).encode(), vec.ctypes.data,
                                                vec.shape[0], fm))

    def flush(self) -> None:
        hip.check(self.lib.sg_writer_flush(self.handle))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.sg_writer_flush(self.handle)
            self.lib.sg_writer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_label_files(output_root: str, result: SceneResult, formats=("txt", "npy")) -> List[str]:
    """The a16 file side (model.py:536-547): `<name>.txt` one '%d\\n' per raw vertex, and `<name>.npy`."""
    lib = hip.lib()
    os.makedirs(output_root, exist_ok=True)
    written = []
    for i in range(result.n_vectors):
        vec = np.ascontiguousarray(result.labels[i])
        base = os.path.join(output_root, hip.LABEL_NAMES[i])
        if "txt" in formats:
            hip.check(lib.sg_write_label_txt((base + ".txt").encode(), vec.ctypes.data, vec.shape[0]))
            written.append(base + ".txt")
        if "npy" in formats:
            hip.check(lib.sg_write_label_npy((base + ".npy").encode(), vec.ctypes.data, vec.shape[0]))
            written.append(base + ".npy")
    return written


class _SceneLoss(torch.autograd.Function):
    """loss [1,2] of one scene as a function of the module's parameters: backward = csrc/trainer.cpp's chain.  The upstream
    gradient of loss[0,0] (= 1 / loss_num in train.py:166) is the scale of the whole backward; loss[0,1] (the instance count)
    carries none."""

    @staticmethod
    def forward(ctx, tr, mask, loss, module, *params):
        ctx.tr, ctx.mask, ctx.module = tr, mask, module
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.token = tr.steps_forward = getattr(tr, "steps_forward", 0) + 1
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        tr = ctx.tr
        if ctx.token != tr.steps_forward:
            raise RuntimeError("backward through a scene whose record was overwritten by a later forward of the same SegModel")
        flat = tr.backward(ctx.mask, float(gout[0, 0]))
        if ctx.module is not None:
            ctx.module._update_running_stats(tr)
        grads, off = [], 0
        for shp in ctx.shapes:
            n = int(np.prod(shp))
            grads.append(flat[off:off + n].reshape(shp).clone())
            off += n
        return (None, None, None, None) + tuple(grads)


class SegModel(nn.Module):
    """Drop-in for the reference `SegModel` (model.py:658-932).

    With neither infer flag set, `forward` runs the whole ins_infer forward (pseudo labels are exported under
    `epoch_<n>/` like the reference does) and then the train-mode tail of model.py:900-932 on HIP, returning
    `(loss[1,2], IoU_sem, IoU_ins, acc)`; `loss` carries an autograd node whose backward is the hand-written HIP chain
    (csrc/trainer.cpp), so `loss.backward()` leaves `.grad` on every parameter like the reference's autograd does.
    """

    def __init__(self, exp_name='exp', cuda=True, visualize=False, sem_infer=False, ins_infer=False,
                 data_root: Optional[str] = None, out_formats=("txt", "npy"), label_style: str = "manual"):
        super().__init__()
        self.exp_name = exp_name
        self.cuda_flag = cuda
        self.visualize = visualize
        self.sem_infer = sem_infer
        self.ins_infer = ins_infer
        self.out_formats = tuple(out_formats)
        self.label_style = label_style
        self.root = data_root if data_root is not None else "."
        self.data_root = os.path.join(self.root, 'dataset', 'scannet')
        scene_list_path = os.path.join(self.data_root, 'scannetv2_train.txt')
        self.scene_list: List[str] = []
        if os.path.exists(scene_list_path):
            with open(scene_list_path, 'r') as f:
                self.scene_list = f.readlines()           # entries keep their '\n' like the reference (model.py:671-672)
        self.epoch = '0'
        self.mlp_1 = MLP1()
        self.mlp_2 = MLP2()
        self.gcn_2 = GCN(dim_in=192, dim_out=192)
        self.mlp_3 = MLP3()
        self.gcn_3 = GCN(dim_in=256, dim_out=256)
        self.classifier = Classifier(dim_in=256, dim_out=40)
        self._pipe: Optional[Pipeline] = None
        self._pipe_key = None
        self._scene_cache: Dict[str, DeviceScene] = {}
        self._lock = threading.Lock()
        self._writer: Optional[AsyncLabelWriter] = None
        self.async_write = True          # label files are written by native threads; flush() waits for them
        self.last_result: Optional[SceneResult] = None
        self.last_tail = None
        self._trainer = None
        self.last_logits = None
        # classifier Dropout(p=0.5) in train mode: "random" (this build's generator), "pinned" (the counter-based mask of the golden
        # capture: parity with the reference is stated with the same mask on both sides), None, or a callable K -> [K,128] mask
        self.dropout_keep = "random"

    # -- parameters -> C ABI ----------------------------------------------------------------------
    def export_weights(self) -> Dict[str, np.ndarray]:
        return _weights.from_state_dict(self.state_dict())

    def load_weights(self, w: Dict[str, np.ndarray]) -> None:
        self.load_state_dict(_weights.to_state_dict(w, prefix=""), strict=False)

    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def pipeline_for(self, sc: DeviceScene) -> Pipeline:
        key = (self._weights_key(), str(sc.device))
        if self._pipe is None or self._pipe_key != key or not self._pipe.fits(sc):
            if self._pipe is not None:
                self._pipe.close()
            caps = (sc.N, sc.S, sc.E0, sc.V) if self._pipe is None or self._pipe_key != key else \
                tuple(max(a, b) for a, b in zip(self._pipe.caps, (sc.N, sc.S, sc.E0, sc.V)))
            self._pipe = Pipeline(self.export_weights(), *caps, stream=None, device=sc.device)
            self._pipe_key = key
        return self._pipe

    def mode(self) -> int:
        if self.sem_infer:
            return hip.MODE_SEM_INFER
        return hip.MODE_INS_INFER

    def output_root(self, scene_name: str) -> str:
        tag = self.epoch if self.epoch in ['sem_infer', 'ins_infer'] else 'epoch_' + self.epoch   # model.py:688-691
        return os.path.join(self.root, 'results', self.exp_name, scene_name, tag)

    # -- forward ------------------------------------------------------------------------------------
    def forward_scene(self, sc: DeviceScene, write: bool = True) -> SceneResult:
        """Hot path on a staged scene.  Returns the SceneResult; writes the label files if `write`."""
        with self._lock:
            if self.sem_infer or self.ins_infer:
                res = self.pipeline_for(sc).forward(sc, self.mode())
            else:                                            # train mode: the same forward, recorded for the backward (csrc/trainer.cpp)
                res = self.trainer_for(sc).forward(sc)
            if write:
                if self.async_write:
                    if self._writer is None:
                        self._writer = AsyncLabelWriter()
                    self._writer.submit(self.output_root(sc.name), res, self.out_formats)
                else:
                    write_label_files(self.output_root(sc.name), res, self.out_formats)
            self.last_result = res
        return res

    def train_tail(self, sc: DeviceScene, res: SceneResult):
        """The train-mode tail as a standalone operator (functional.TrainTail) from a Feat_5 tap (`Pipeline.forward(want_feat5=True)`);
        the training step proper goes through `trainer_for` / `forward`."""
        if res.feat5 is None:
            raise RuntimeError("train_tail needs a forward made with want_feat5=True (Feat_5 tap)")
        K = int(np.unique(res.ins5).shape[0])
        keep = self.dropout_keep(K) if callable(self.dropout_keep) else self._trainer_mask(K)
        cls = {k[len("classifier."):]: v for k, v in self.state_dict().items() if k.startswith("classifier.")}
        return _F.TrainTail(torch.from_numpy(res.feat5).to(sc.device), res.ins5, res.sem5, cls, keep)

    def _trainer_mask(self, K: int):
        if self.dropout_keep is None:
            return None
        if isinstance(self.dropout_keep, str) and self.dropout_keep == "pinned":
            from .synthetic import uniform01
            return np.where(uniform01(97, int(K), int(K) * 128).reshape(int(K), 128) < 0.5, 2.0, 0.0).astype(np.float32)
        if isinstance(self.dropout_keep, str):
            return (np.random.default_rng().random((K, 128)) >= 0.5).astype(np.float32) * 2.0
        return np.asarray(self.dropout_keep, np.float32)

    def flush(self) -> None:
        """Wait until every label file submitted so far is on disk (raises on the first I/O error)."""
        if self._writer is not None:
            self._writer.flush()

    def forward(self, data, weak_label, info):
        data, weak_label, info = data[0], weak_label[0], info[0]
        if not data.is_cuda:
            raise RuntimeError("seggroup_amd.SegModel.forward needs CUDA/HIP tensors: there is no CPU path "
                               "(use oracle/cpu_ref.py only for testing)")
        scene_name = self.scene_list[int(info)][:-1]
        sc = self._scene_cache.get(scene_name)
        if sc is None:
            sc = DeviceScene.from_reference_tree(scene_name, data=data, weak_label=weak_label, root=self.root,
                                                 label_style=self.label_style, device=data.device)
            self._scene_cache = {scene_name: sc}          # keep one scene resident, like the reference's per-step load
        res = self.forward_scene(sc, write=True)
        if res.stalled:
            print('%s: a <5-point cluster could not be merged (the reference loops forever here, model.py:228-239); sweep cut short'
                  % scene_name, flush=True)
        dev = data.device
        out = (torch.from_numpy(res.iou_sem).to(dev), torch.from_numpy(res.iou_ins).to(dev), torch.from_numpy(res.acc).to(dev))
        if self.sem_infer or self.ins_infer:
            return out
        # train mode (model.py:900-932): the loss carries an autograd node whose backward runs the whole HIP
        # backward chain and hands every parameter its gradient, so the reference's loop body (train.py:160-168) --
        #     loss = loss_raw[:, 0].sum() / loss_raw[:, 1].sum();  optimizer.zero_grad();  loss.backward();  optimizer.step()
        # -- works unchanged with torch.optim and DistributedDataParallel around this module.  (seggroup_amd/train.py drives the
        # same kernels through flat vectors and one all-reduce instead.)
        return (self._train_loss(sc),) + out

    def trainer_for(self, sc: DeviceScene):
        """The `Trainer` behind train mode, holding THIS module's current parameter values (flat copy, 0.59 MB per forward)."""
        from . import trainer as _T
        tr = self._trainer
        if tr is None or not tr.fits(sc) or tr.device != sc.device:
            if tr is not None:
                tr.close()
            caps = (sc.N, sc.S, sc.E0, sc.V) if tr is None else tuple(max(a, b) for a, b in zip(tr.caps, (sc.N, sc.S, sc.E0, sc.V)))
            tr = self._trainer = _T.Trainer({k: v for k, v in self.state_dict().items()}, caps, device=sc.device)
        with torch.no_grad():
            tr.params.copy_(torch.cat([p.detach().reshape(-1).float() for _, p in self.named_parameters()]))
        return tr

    def _train_loss(self, sc: DeviceScene):
        params = [p for _, p in self.named_parameters()]
        with self._lock:
            tr = self._trainer                               # forward_scene just ran this scene through it
            keep = self.dropout_keep(tr.K) if callable(self.dropout_keep) else self.dropout_keep
            mask = tr.dropout_mask(keep)
            loss = torch.from_numpy(tr.loss(mask, want_logits=True)).to(sc.device)
            self.last_logits = tr.logits
        return _SceneLoss.apply(tr, mask, loss, self if self.training else None, *params)

    def _update_running_stats(self, tr) -> None:
        """BatchNorm's running statistics (momentum 0.1).  The batch statistics of the three EdgeConv / MLP1 BatchNorms fall out of
        the backward kernels, so the buffers move when `loss.backward()` has run -- not, as in the reference, during forward."""
        with torch.no_grad():
            bufs = dict(self.named_buffers())
            for k in tr.buffers:
                tr.buffers[k] = bufs[k].detach().cpu().clone()
            tr.update_running_stats()
            for k, b in bufs.items():
                src = tr.buffers.get(k.replace(".conv1.1.", ".bn1.").replace(".conv2.1.", ".bn2."))
                if src is not None:
                    b.copy_(src.to(b.device))
