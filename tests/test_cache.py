"""Packed scene cache: the pack holds exactly the staged arrays `DeviceScene` uploads, is rebuilt
when a source file of the reference's tree changes, and rejects foreign files.  CPU only (no upload here)."""
import os
import time

import numpy as np
import pytest

from conftest import make_fixture_scene


def _tree(tmp_path, golden_index, name="tiny_dup_4k"):
    from seggroup_amd import synthetic
    scene = make_fixture_scene(golden_index, name)
    synthetic.write_reference_tree(str(tmp_path), [scene])
    return scene


def test_pack_roundtrip_equals_staged_arrays(tmp_path, golden_index):
    from seggroup_amd import cache
    scene = _tree(tmp_path, golden_index)
    want = cache.stage_arrays(scene.data, scene.weak_label, scene.seg, scene.adj, scene.unmap, scene.gt)
    path = cache.pack_scene(str(tmp_path), scene.name)
    assert path == cache.pack_path(str(tmp_path), scene.name) and os.path.exists(path)
    got = cache.read_pack(path)
    assert got["name"] == scene.name and got["N"] == scene.data.shape[0] and got["V"] == scene.unmap.shape[0]
    assert got["S"] == int(scene.seg.max()) + 1 and got["E0"] == scene.adj.shape[0]
    for k in cache.ARRAYS:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert np.array_equal(got[k], want[k]), k
        if k != "adj":                                                    # (adj is stored as int32 and widened on the way out: a copy)
            assert got[k].ctypes.data % 64 == 0 or got[k].size == 0, k    # 64-byte aligned inside the page-aligned map
    import json
    import struct
    with open(path, "rb") as f:
        f.read(8)
        hdr = json.loads(f.read(struct.unpack("<I", f.read(4))[0]).decode())
    assert hdr["arrays"]["adj"][0] == "<i4" and hdr["arrays"]["adj"][2] % 64 == 0     # half the bytes of the reference's int64 rows
    # CSR invariants the kernels rely on: ascending point index inside each segment, first point ascending
    off, pts = got["seg_off"], got["seg_points"]
    assert off[0] == 0 and off[-1] == got["N"] and (np.diff(off) == got["seg_size"]).all()
    assert (pts[off[:-1]] == got["seg_first"]).all()
    inner = np.ones(got["N"], bool)
    inner[off[:-1]] = False
    assert (np.diff(pts)[inner[1:]] > 0).all()


def test_pack_is_reused_then_rebuilt_when_a_source_is_newer(tmp_path, golden_index):
    from seggroup_amd import cache
    scene = _tree(tmp_path, golden_index, "tiny_4k")
    path = cache.pack_scene(str(tmp_path), scene.name)
    m0 = os.stat(path).st_mtime_ns
    assert cache.pack_scene(str(tmp_path), scene.name) == path and os.stat(path).st_mtime_ns == m0     # reused
    src = cache.source_files(str(tmp_path), scene.name)[3]                                             # the seg.json
    future = time.time() + 5
    os.utime(src, (future, future))
    cache.pack_scene(str(tmp_path), scene.name)
    assert os.stat(path).st_mtime_ns != m0                                                             # rebuilt
    m1 = os.stat(path).st_mtime_ns
    cache.pack_scene(str(tmp_path), scene.name, force=True)
    assert os.stat(path).st_mtime_ns != m1


def test_label_styles_get_separate_packs_and_foreign_files_are_rejected(tmp_path, golden_index):
    from seggroup_amd import cache
    assert cache.pack_path("r", "s", "manual") != cache.pack_path("r", "s", "rand_inside")
    bad = tmp_path / "x.sgpack"
    bad.write_bytes(b"NOTAPACK" + b"\0" * 64)
    with pytest.raises(ValueError):
        cache.read_pack(str(bad))


def test_stage_arrays_rejects_inconsistent_input(golden_index):
    from seggroup_amd import cache
    scene = make_fixture_scene(golden_index, "tiny_4k")
    with pytest.raises(ValueError):
        cache.stage_arrays(scene.data[:, :5], scene.weak_label, scene.seg, scene.adj, scene.unmap, scene.gt)
    seg = scene.seg.copy()
    a, b = seg == 0, seg == 1
    seg[a], seg[b] = 1, 0                     # segment numbers no longer ascend with first points
    with pytest.raises(ValueError):
        cache.stage_arrays(scene.data, scene.weak_label, seg, scene.adj, scene.unmap, scene.gt)


def test_missing_packs_are_built_in_worker_processes(tmp_path, golden_index):
    """cache.build_missing: stale / missing packs only, through the process pool once there are enough of them."""
    from seggroup_amd import cache, synthetic
    base = make_fixture_scene(golden_index, "tiny_4k")
    scenes = [synthetic.Scene(f"scene{i:04d}_00", base.data, base.weak_label, base.seg, base.adj, base.unmap, base.gt) for i in range(10)]
    root = str(tmp_path)
    synthetic.write_reference_tree(root, scenes)
    names = [s.name for s in scenes]
    assert not cache.is_current(root, names[0])
    cache.pack_scene(root, names[0])                                  # one is already there
    assert cache.is_current(root, names[0])
    assert cache.build_missing(root, names, workers=4) == 9          # >= 8 to do: worker processes
    assert all(cache.is_current(root, n) for n in names) and cache.build_missing(root, names, workers=4) == 0
    want = cache.stage_arrays(base.data, base.weak_label, base.seg, base.adj, base.unmap, base.gt)
    got = cache.read_pack(cache.pack_path(root, names[7]))
    assert got["name"] == names[7] and all(np.array_equal(got[k], want[k]) for k in cache.ARRAYS)
    src = cache.source_files(root, names[3])[0]
    future = time.time() + 5
    os.utime(src, (future, future))
    assert not cache.is_current(root, names[3]) and cache.build_missing(root, names, workers=4) == 1     # few: built inline


# ---- the label files as the three downstream trainers read them ---------------------------------

def _read_like_consumers(path, as_type):
    """pointgroup/dataset/scannetv2/prepare_data_inst2.py:41-53, minkowski/lib/datasets/preprocessing/scannet2.py:29-34,
    kpconv/datasets/Scannet2.py:148-153 all do: readlines(), drop the last character of every line, np.array().astype()."""
    with open(path, "r") as f:
        lines = f.readlines()
    return np.array([ln[:-1] for ln in lines]).astype(as_type)


@pytest.mark.parametrize("use_pool", [False, True])
def test_label_files_are_what_the_reference_writes_and_its_consumers_parse(tmp_path, use_pool):
    import ctypes as C
    from seggroup_amd import hip
    lib = hip.lib()
    rng = np.random.default_rng(5)
    vecs = [np.concatenate([[-1, 0, 1, 9, 10, 40, -100, 2**31 - 1, -(2**31) + 1], rng.integers(-1, 150000, 20000)]).astype(np.int32),
            np.zeros(0, np.int32), np.array([7], np.int32)]
    paths = []
    if use_pool:
        w = lib.sg_writer_create(3, 4)
        assert w
        for i, v in enumerate(vecs):
            base = str(tmp_path / f"final.{i}")
            hip.check(lib.sg_writer_submit(w, base.encode(), v.ctypes.data if v.size else None, v.size, 3))
            paths.append(base)
        hip.check(lib.sg_writer_flush(w))
        lib.sg_writer_destroy(w)
    else:
        for i, v in enumerate(vecs):
            base = str(tmp_path / f"final.{i}")
            hip.check(lib.sg_write_label_txt((base + ".txt").encode(), v.ctypes.data if v.size else None, v.size))
            hip.check(lib.sg_write_label_npy((base + ".npy").encode(), v.ctypes.data if v.size else None, v.size))
            paths.append(base)
    for base, v in zip(paths, vecs):
        # byte-for-byte what model.py:541-545 writes: '%d\n' per raw vertex
        assert open(base + ".txt", "rb").read() == "".join("%d\n" % x for x in v.tolist()).encode()
        got = np.load(base + ".npy")
        assert got.dtype == np.int32 and np.array_equal(got, v)
        if v.size:
            assert np.array_equal(_read_like_consumers(base + ".txt", int), v.astype(np.int64))
            assert np.array_equal(_read_like_consumers(base + ".txt", float), v.astype(np.float64))
            assert np.array_equal(_read_like_consumers(base + ".txt", "float32"), v.astype(np.float32))
