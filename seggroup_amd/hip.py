"""ctypes binding of libseggroup_hip.so (C ABI declared in include/seggroup_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or no HIP device is
visible when a device entry point is needed, this module raises -- it never routes to the oracle.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEGGROUP_HIP_LIB: another build of the same library (a profiling build: `make PROFILE=1`, tools/ec_phases.py)
LIB_PATH = os.environ.get("SEGGROUP_HIP_LIB") or os.path.join(_HERE, "libseggroup_hip.so")
# SEGGROUP_HIP_HOST_LIB: a HOST-ONLY build of the library (`make -C seggroup_amd/csrc asan`: grouping engine, writers,
# parsers under ASan/UBSan).  It has no kernels, so only the host entry points bind; everything else raises on use.
HOST_LIB_OVERRIDE = os.environ.get("SEGGROUP_HIP_HOST_LIB")

SG_OK, SG_EINVAL, SG_EHIP, SG_ENOMEM, SG_ESTALL, SG_EUNSUP = 0, -1, -2, -3, -4, -5
MODE_INS_INFER, MODE_SEM_INFER = 0, 1
NUM_LABEL_VECTORS = 14
LABEL_NAMES = [f"layer_{l}.{k}" for l in (1, 2, 3, 4) for k in ("seg", "ins", "sem")] + ["final.ins", "final.sem"]

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class SgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libseggroup_hip error {code}: {msg}")
        self.code = code


class Weights(C.Structure):
    _fields_ = [(n, vp) for n in ("mlp1_w", "mlp1_g", "mlp1_b", "mlp2_w", "mlp2_g", "mlp2_b", "gcn2_w",
                                  "mlp3_w1", "mlp3_g1", "mlp3_b1", "mlp3_w2", "mlp3_g2", "mlp3_b2", "gcn3_w")]


class Scene(C.Structure):
    _fields_ = [("N", C.c_int), ("S", C.c_int), ("E0", C.c_int), ("V", C.c_int),
                ("d_data", vp), ("d_adj", vp), ("d_seg_of_point", vp), ("d_seg_points", vp), ("d_seg_off", vp),
                ("d_unmap", vp), ("d_gt", vp),
                ("h_seg_first", vp), ("h_seg_size", vp), ("h_seg_ins", vp), ("h_seg_sem", vp), ("h_seg_of_vertex", vp)]


class Result(C.Structure):
    _fields_ = [("h_labels", vp), ("iou_sem", C.c_float * 80), ("iou_ins", C.c_float * 80), ("acc", C.c_float * 4),
                ("trace", C.c_int32 * 5), ("stalled", C.c_int32), ("used_fallback", C.c_int32), ("h_tables", vp)]


class Classifier(C.Structure):
    _fields_ = [(n, vp) for n in ("w1", "gamma", "beta", "w2", "b2")]


class Debug(C.Structure):
    _fields_ = [("d_samples1", vp), ("d_feat1", vp), ("d_pointfeat", vp * 2), ("d_knn", vp * 2), ("d_members", vp * 2),
                ("h_gcn", vp * 2), ("h_dist", vp * 3), ("h_adj", vp * 4), ("n_adj", C.c_int32 * 4),
                ("h_feat5", vp), ("h_ins5", vp), ("h_sem5", vp), ("n5", C.c_int32), ("tape", vp)]


# name -> (restype, argtypes); every symbol declared in include/seggroup_hip.h
_I, _Z = C.c_int, C.c_size_t
SIGNATURES = {
    "sg_last_error": (C.c_char_p, []),
    "sg_version": (_I, []),
    "sg_device_count": (_I, []),
    "sg_selftest_wave_ops": (_I, [C.POINTER(C.c_int), vp]),
    "sg_selftest_list_insert": (_I, [C.POINTER(C.c_int), vp]),
    "sg_contract_ws_bytes": (_Z, [_I]),
    "sg_contract_point_edges": (_I, [vp, _I, vp, _I, _I, vp, _I, vp, vp, _Z, vp]),
    "sg_gather_members": (_I, [vp, vp, _I, vp, vp, vp, vp, vp, vp, vp, vp]),
    "sg_fps_ws_bytes": (_Z, [_I]),
    "sg_fps_sample": (_I, [vp, _I, _I, vp, vp, _I, _I, _I, _I, vp, vp, vp, _Z, vp]),
    "sg_mlp1_ws_bytes": (_Z, [_I]),
    "sg_mlp1_forward": (_I, [vp, _I, vp, vp, vp, vp, _I, vp, _Z, vp]),
    "sg_edge_distance": (_I, [vp, _I, _I, vp, _I, vp, vp]),
    "sg_group_max_rows": (_I, [vp, _I, _I, vp, vp, _I, vp, _I, vp]),
    "sg_segment_max": (_I, [vp, _I, _I, vp, vp, _I, _I, vp]),
    "sg_center_ws_bytes": (_Z, [_I, _I]),
    "sg_center_clusters": (_I, [vp, _I, vp, vp, _I, vp, vp, vp, _I, vp, vp, vp, vp, _Z, vp]),
    "sg_cluster_knn": (_I, [vp, _I, vp, vp, vp, vp, _I, _I, _I, vp, vp]),
    "sg_segment_boxes": (_I, [vp, vp, vp, _I, vp, vp]),
    "sg_cluster_knn_pruned": (_I, [vp, _I, vp, vp, vp, vp, _I, vp, vp, vp, vp, vp, vp, _I, _I, vp, vp]),
    "sg_segment_sort_ws_bytes": (_Z, [_I]),
    "sg_knn_operands": (_I, [vp, vp, vp, vp, _I, vp, vp, vp, vp, vp]),
    "sg_cluster_knn_sorted": (_I, [vp, vp, _I, vp, vp, vp, vp, _I, vp, vp, vp, vp, vp, vp, vp, vp, _I, _I, vp, vp]),
    "sg_segment_sort_boxes": (_I, [vp, _I, vp, vp, vp, _I, vp, _I, vp, vp, vp, vp, vp, _Z, vp]),
    "sg_layer_layout": (_I, [vp, _I, vp, vp, vp, _I, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "sg_cluster_knn_sorted_w": (_I, [vp, vp, _I, vp, vp, vp, vp, _I, vp, vp, vp, vp, vp, vp, vp, vp, _I, _I, _I, vp, vp]),
    "sg_knn_seed_points": (_I, [vp, vp, _I, _I, vp, vp]),
    "sg_cluster_knn_seeded": (_I, [vp, vp, _I, vp, vp, vp, vp, _I, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, _I, _I, vp, vp]),
    "sg_knn_chunk_table": (_I, [vp, vp, vp, vp, vp, _I, vp, vp, vp]),
    "sg_cluster_knn_2pass": (_I, [vp, vp, _I, vp, vp, vp, vp, vp, _I, vp, vp, _I, _I, vp, vp]),
    "sg_nearest_point_ws_bytes": (_Z, [_I]),
    "sg_nearest_point": (_I, [vp, _I, vp, _I, _I, vp, vp, _Z, vp]),
    "sg_prep_sample_ws_bytes": (_Z, [_I, _I]),
    "sg_prep_sample_points": (_I, [vp, vp, _I, vp, _I, vp, vp, C.POINTER(C.c_int), vp, _Z, vp]),
    "sg_pointcloud_adjacency_ws_bytes": (_Z, [_I, _I]),
    "sg_pointcloud_adjacency": (_I, [vp, _I, _I, _I, vp, C.POINTER(C.c_int), vp, _Z, vp]),
    "sg_mesh_adjacency_ws_bytes": (_Z, [_I]),
    "sg_mesh_adjacency": (_I, [vp, _I, vp, _I, vp, C.POINTER(C.c_int), vp, C.POINTER(C.c_int), vp, _Z, vp]),
    "sg_segment_lists_ws_bytes": (_Z, [_I, _I]),
    "sg_segment_lists": (_I, [vp, _I, vp, _I, vp, vp, vp, C.POINTER(C.c_int), vp, _Z, vp]),
    "sg_write_seg_json": (_I, [C.c_char_p, vp, vp, _I, _I]),
    "sg_train_tail_ws_bytes": (_Z, [_I, _I]),
    "sg_train_tail_forward": (_I, [vp, _I, vp, _I, vp, vp, vp, vp, vp, vp, _Z, vp]),
    "sg_train_tail_backward": (_I, [_I, _I, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp, _Z, vp]),
    "sg_group_mean_rows": (_I, [vp, _I, _I, vp, vp, _I, vp, _I, vp]),
    "sg_fps_general_ws_bytes": (C.c_size_t, [_I]),
    "sg_fps_general": (_I, [vp, _I, _I, _I, _I, _I, vp, vp, vp, C.c_size_t, vp]),
    "sg_knn_general": (_I, [vp, _I, _I, _I, _I, vp, vp]),
    "sg_group_max_rows_backward": (_I, [vp, _I, _I, vp, vp, _I, vp, _I, vp, _I, vp]),
    "sg_segment_max_backward_ws_bytes": (_Z, [_I, _I]),
    "sg_segment_max_backward": (_I, [vp, _I, _I, vp, _I, vp, _I, vp, vp, _Z, vp]),
    "sg_gcn_backward_ws_bytes": (_Z, [_I, _I, _I]),
    "sg_gcn_backward": (_I, [vp, _I, _I, vp, _I, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, _Z, vp]),
    "sg_cross_entropy_forward": (_I, [vp, _I, _I, vp, _I, vp, vp, vp]),
    "sg_cross_entropy_backward": (_I, [vp, _I, _I, vp, _I, C.c_float, vp, vp]),
    "sg_train_tail_bn_stats": (_I, [vp, _Z, _I, vp, vp]),
    "sg_param_slot": (_I, [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sg_trainer_create": (vp, [_I, _I, _I, _I, vp, vp, vp]),
    "sg_trainer_destroy": (None, [vp]),
    "sg_trainer_device_bytes": (_Z, [vp]),
    "sg_trainer_forward": (_I, [vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sg_trainer_loss": (_I, [vp, vp, C.POINTER(C.c_float), vp]),
    "sg_trainer_backward": (_I, [vp, vp, C.c_float]),
    "sg_trainer_bn_stats": (_I, [vp, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "sg_optimizer_sgd": (_I, [vp, vp, vp, _I, C.c_float, C.c_float, C.c_float, _I, vp]),
    "sg_optimizer_adam": (_I, [vp, vp, vp, vp, _I, C.c_float, C.c_float, _I, vp]),
    "sg_mlp1_backward_ws_bytes": (_Z, [_I]),
    "sg_mlp1_backward": (_I, [vp, _I, vp, vp, vp, vp, _I, vp, vp, vp, vp, vp, _Z, vp]),
    "sg_edgeconv_backward_ws_bytes": (_Z, [_I]),
    "sg_edgeconv_backward": (_I, [vp, vp, _I, _I, _I] + [vp] * 15 + [vp, _Z, vp]),
    "sg_parse_seg_json": (_I, [C.c_char_p, _I, vp]),
    "sg_stage_segments": (_I, [vp, _I, _I, vp, vp, vp, vp]),
    "sg_edgeconv_ws_bytes": (_Z, [_I]),
    "sg_edgeconv_forward": (_I, [vp, vp, _I, _I, _I, vp, vp, vp, vp, vp, vp, vp, vp, _Z, vp]),
    "sg_edgeconv_forward_r": (_I, [vp, vp, _I, _I, _I, vp, vp, vp, vp, vp, vp, vp, vp, _Z, vp, vp]),
    "sg_edgeconv_forward_x": (_I, [vp, vp, _I, _I, _I, vp, vp, vp, vp, vp, vp, vp, vp, _Z, vp, C.c_uint, vp]),
    "sg_edge_range": (_I, [vp, _I, vp, vp]),
    "sg_gcn_ws_bytes": (_Z, [_I, _I, _I]),
    "sg_gcn_forward": (_I, [vp, _I, _I, vp, _I, vp, vp, vp, vp, C.c_float, vp, vp, _Z, vp]),
    "sg_export_labels": (_I, [vp, _I, vp, _I, vp, _I, _I, vp, vp]),
    "sg_eval_ws_bytes": (_Z, [_I]),
    "sg_evaluate": (_I, [vp, vp, vp, _I, _I, vp, vp, vp, vp, _Z, vp]),
    "sg_partition_create": (vp, [_I, vp, vp, vp, vp]),
    "sg_partition_destroy": (None, [vp]),
    "sg_partition_num_clusters": (_I, [vp]),
    "sg_partition_union": (_I, [vp, _I, _I]),
    "sg_partition_find": (_I, [vp, _I]),
    "sg_partition_label": (_I, [vp, _I, vp, vp, vp]),
    "sg_partition_layer": (_I, [vp, vp, vp, vp, vp, vp, vp]),
    "sg_partition_group_nearby": (_I, [vp, vp, _I, vp, vp, _I, C.c_float, vp]),
    "sg_partition_contract": (_I, [vp, vp, vp, _I, vp, vp]),
    "sg_partition_group_unlabeled": (_I, [vp, vp, vp, vp, _I, vp, vp]),
    "sg_partition_unlabeled_fallback": (_I, [vp, vp, _I, vp, _I]),
    "sg_partition_export_tables": (_I, [vp, vp, vp, vp]),
    "sg_pipeline_create": (vp, [_I, _I, _I, _I, vp, vp]),
    "sg_pipeline_destroy": (None, [vp]),
    "sg_pipeline_device_bytes": (_Z, [vp]),
    "sg_pipeline_forward": (_I, [vp, vp, _I, vp, vp]),
    "sg_batch_forward": (_I, [vp, _I, vp, _I, _I, vp, vp, vp, vp, _I]),
    "sg_engine_create": (vp, [_I, _I, _I, _I, vp, _I, _I]),
    "sg_engine_destroy": (None, [vp]),
    "sg_engine_submit": (_I, [vp, vp, _I, _I, vp, vp, vp, _I]),
    "sg_engine_wait": (_I, [vp, _I]),
    "sg_engine_set_timing": (_I, [vp, _I]),
    "sg_engine_set_label_transfer": (_I, [vp, _I]),
    "sg_engine_set_knn_variant": (_I, [vp, _I]),
    "sg_engine_stage_times": (C.c_longlong, [vp, vp, _I, _I]),
    "sg_engine_device_bytes": (_Z, [vp]),
    "sg_engine_profile": (_I, [vp, vp, _I, _I]),
    "sg_pipeline_stage_times": (_I, [vp, vp, _I]),
    "sg_pipeline_set_timing": (_I, [vp, _I]),
    "sg_pipeline_set_knn_variant": (_I, [vp, _I]),
    "sg_pipeline_stage_name": (C.c_char_p, [_I]),
    "sg_write_label_txt": (_I, [C.c_char_p, vp, _I]),
    "sg_write_label_npy": (_I, [C.c_char_p, vp, _I]),
    "sg_writer_create": (vp, [_I, _I]),
    "sg_writer_submit": (_I, [vp, C.c_char_p, vp, _I, _I]),
    "sg_loader_create": (vp, [_I, _I, _Z]),
    "sg_loader_create_sized": (vp, [_I, _I, _Z, _Z]),
    "sg_loader_set_copy_limit": (_I, [vp, _I]),
    "sg_loader_submit": (_I, [vp, C.c_char_p]),
    "sg_loader_wait": (_I, [vp, _I, vp, C.POINTER(C.c_int), C.c_char_p, _I]),
    "sg_loader_release": (_I, [vp, _I]),
    "sg_loader_destroy": (None, [vp]),
    "sg_writer_submit_scene": (_I, [vp, C.c_char_p, vp, _I, _I, _I, C.c_longlong]),
    "sg_writer_submit_scene_tables": (_I, [vp, C.c_char_p, vp, _I, vp, _I, _I, _I, C.c_longlong]),
    "sg_expand_labels": (_I, [vp, _I, _I, vp, _I, vp]),
    "sg_pack_build": (_I, [vp, C.c_char_p, C.c_char_p]),
    "sg_pack_build_many": (_I, [vp, vp, vp, _I, _I, vp]),
    "sg_writer_wait_tag": (_I, [vp, C.c_longlong]),
    "sg_writer_flush": (_I, [vp]),
    "sg_writer_destroy": (None, [vp]),
}

_lib: Optional[C.CDLL] = None
_runtime = None


def _load_hip_runtime():
    """Make ONE HIP runtime visible (RTLD_GLOBAL) before libseggroup_hip.so binds its hip* symbols.

    PyTorch-ROCm bundles its own libamdhip64.so; the library must use that same runtime (streams and
    events are runtime-local objects, and a second runtime initialised in the process does not find
    the GPU).  Without torch (plain C++/ctypes host) the system ROCm runtime is used.
    """
    global _runtime
    if _runtime is not None:
        return _runtime
    cands = []
    import sys
    if "torch" in sys.modules or not os.environ.get("SEGGROUP_HOST_ONLY"):
        try:
            import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
            cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        except ImportError:
            pass
    # SEGGROUP_HOST_ONLY=1 (pack-building worker processes: parsers and writers only): do not pay ~2 s for importing torch
    # just to find its copy of the runtime; a process that will touch the GPU through torch must not set it
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    err = None
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            _runtime = C.CDLL(c, mode=C.RTLD_GLOBAL)
            return _runtime
        except OSError as e:  # pragma: no cover
            err = e
    raise RuntimeError(f"no HIP runtime (libamdhip64.so) could be loaded: {err}")


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        path = HOST_LIB_OVERRIDE or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C seggroup_amd/csrc`).  The SegGroup hot path has no CPU fallback.")
        _load_hip_runtime()
        l = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if HOST_LIB_OVERRIDE:              # host-only sanitizer build: device entry points are absent by design
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_roctx = None


@contextlib.contextmanager
def roctx_range(name: str):
    """A roctx range around a block of host code (`bench.py --profile`, SG_ROCTX=1); does nothing otherwise.  The engine's group threads push
    their own ranges per phase and stage (csrc/engine.cpp); `rocprofv3 --kernel-trace --marker-trace` shows both."""
    global _roctx
    if not os.environ.get("SG_ROCTX"):
        yield
        return
    if _roctx is None:
        _roctx = False
        for cand in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                _roctx = C.CDLL(cand, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
    if not _roctx:
        yield
        return
    _roctx.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        _roctx.roctxRangePop()


def check(rc: int) -> int:
    if rc < 0:
        raise SgError(rc, lib().sg_last_error().decode("utf-8", "replace"))
    return rc


def require_device() -> None:
    if lib().sg_device_count() <= 0:
        raise RuntimeError("no HIP device visible: the SegGroup hot path runs on MI355X only (no CPU fallback)")


def ptr(t) -> int:
    """Device / host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
