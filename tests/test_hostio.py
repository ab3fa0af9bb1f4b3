"""Native readers of the reference's on-disk inputs: `sg_parse_seg_json` against `json.load` +
`seg_from_lists` (the Python restatement of model.py:713-721's reading of the file)."""
import json
import os

import numpy as np
import pytest


def _write(tmp_path, lists, name="s.seg.json", raw=None):
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        if raw is not None:
            f.write(raw)
        else:
            json.dump(lists, f)
    return p


def test_seg_json_parser_matches_json_load(sg_lib, tmp_path):
    from seggroup_amd import synthetic
    from seggroup_amd.scene import seg_from_file, seg_from_lists
    for n, s, seed in ((2000, 20, 3), (5000, 200, 4), (700, 7, 5)):
        sc = synthetic.make_scene(n, s, seed, min_seg=2)
        lists = sc.seg_lists()
        p = _write(tmp_path, lists, f"a{seed}.seg.json")
        got = seg_from_file(p, n)
        assert np.array_equal(got, seg_from_lists(lists, n)) and np.array_equal(got, sc.seg)
    # whitespace / layout variants json.load accepts as well
    lists = [[0, 2], [1], [], [3, 4], []]
    for raw in (json.dumps(lists), json.dumps(lists, separators=(",", ":")), json.dumps(lists, indent=1), " \n" + json.dumps(lists) + "\n"):
        p = _write(tmp_path, None, "b.seg.json", raw=raw)
        assert seg_from_file(p, 5).tolist() == [0, 1, 0, 2, 2]


@pytest.mark.parametrize("raw,n,needle", [
    ("[[0, 1], [], [1]]", 3, "does not start at its own index"),       # list 2 starts at 1
    ("[[0], [1], []]", 3, "does not cover every point"),
    ("[[0, 5], [1]]", 2, "outside"),
    ("[[0, 1], [1]]", 2, "two lists"),
    ("[[0, 1], [}", 2, "expected"),
    ("[[0], [1]] x", 2, "trailing"),
    ("{}", 2, "does not start"),
])
def test_seg_json_parser_rejects_what_the_python_reader_rejects(sg_lib, tmp_path, raw, n, needle):
    from seggroup_amd import hip
    from seggroup_amd.scene import seg_from_file
    p = _write(tmp_path, None, "bad.seg.json", raw=raw)
    with pytest.raises(hip.SgError) as e:
        seg_from_file(p, n)
    assert needle in str(e.value)


def test_missing_file_is_an_error_not_a_crash(sg_lib, tmp_path):
    from seggroup_amd import hip
    from seggroup_amd.scene import seg_from_file
    with pytest.raises(hip.SgError):
        seg_from_file(os.path.join(str(tmp_path), "nope.seg.json"), 4)


def test_writer_pool_scene_jobs_by_reference(sg_lib, tmp_path):
    """sg_writer_submit_scene (model.py:533-547: the 14 files of an export directory): the vectors are written straight out of the
    caller's buffer by one worker (openat + writev), sg_writer_wait_tag returns once every scene up to a tag is on disk, the single-vector
    entry (copying) and the plain file functions give the same bytes; '%d\\n' text and NumPy-readable .npy, V = 0 included."""
    import ctypes as C
    from seggroup_amd import hip
    lib = sg_lib
    w = lib.sg_writer_create(3, 8)
    rng = np.random.default_rng(5)
    scenes = []
    for t, V in enumerate((1, 0, 1000, 77777)):
        lab = np.ascontiguousarray(rng.integers(-1, 3_000_000, (14, V)).astype(np.int32))
        d = tmp_path / f"scene{t}"
        d.mkdir()
        hip.check(lib.sg_writer_submit_scene(w, str(d).encode(), lab.ctypes.data, V, 14 if t % 2 == 0 else 6, 3, t))
        scenes.append((d, lab, 14 if t % 2 == 0 else 6))
    hip.check(lib.sg_writer_wait_tag(w, 1))                                     # scenes 0 and 1 are complete now
    for d, lab, nvec in scenes[:2]:
        assert sorted(os.listdir(d)) == sorted(f"{n}.{e}" for n in hip.LABEL_NAMES[:nvec] for e in ("txt", "npy"))
    hip.check(lib.sg_writer_flush(w))
    for d, lab, nvec in scenes:
        for i, n in enumerate(hip.LABEL_NAMES[:nvec]):
            got = np.load(d / f"{n}.npy")
            assert got.dtype == np.int32 and np.array_equal(got, lab[i])
            assert (d / f"{n}.txt").read_bytes() == "".join("%d\n" % v for v in lab[i]).encode()
        assert len(os.listdir(d)) == 2 * nvec
    # the copying single-vector entry and the plain functions: same bytes
    v = np.ascontiguousarray(scenes[3][1][0])
    hip.check(lib.sg_writer_submit(w, str(tmp_path / "one").encode(), v.ctypes.data, v.shape[0], 3))
    hip.check(lib.sg_writer_flush(w))
    hip.check(lib.sg_write_label_txt(str(tmp_path / "two.txt").encode(), v.ctypes.data, v.shape[0]))
    hip.check(lib.sg_write_label_npy(str(tmp_path / "two.npy").encode(), v.ctypes.data, v.shape[0]))
    for e in ("txt", "npy"):
        assert (tmp_path / f"one.{e}").read_bytes() == (tmp_path / f"two.{e}").read_bytes() == (scenes[3][0] / f"layer_1.seg.{e}").read_bytes()
    # a missing directory is an error reported by flush, not a crash
    hip.check(lib.sg_writer_submit_scene(w, str(tmp_path / "nope").encode(), v.ctypes.data, v.shape[0], 1, 2, 9))
    assert lib.sg_writer_flush(w) < 0
    lib.sg_writer_destroy(w)


# ---- native pack builder (csrc/packbuild.cpp): runs under ASan / UBSan with the tests above --------------------------------------
def _tree(tmp_path, golden_index, names=("tiny_4k", "tiny_dup_4k")):
    from seggroup_amd import synthetic
    scenes = []
    for i, fx in enumerate(names):
        e = golden_index[fx]
        scenes.append(synthetic.make_scene(e["n"], e["s"], e["seed"], name=f"scene{i:04d}_00", **e["kw"]))
    scenes.append(synthetic.make_scene(2400, 300, 41001, name="scene0009_00", min_seg=1))       # one-point segments
    root = str(tmp_path)
    synthetic.write_reference_tree(root, scenes)
    return root, [s.name for s in scenes]


def _native_build(sg_lib, root, names, threads=2, style="manual"):
    import ctypes as C
    from seggroup_amd import cache
    srcs, outs = [], []
    for n in names:
        srcs += cache.source_files(root, n, style)
        out = cache.pack_path(root, n, style) + ".native"
        os.makedirs(os.path.dirname(out), exist_ok=True)
        outs.append(out)
    a_src = (C.c_char_p * len(srcs))(*[p.encode() for p in srcs])
    a_names = (C.c_char_p * len(names))(*[n.encode() for n in names])
    a_out = (C.c_char_p * len(names))(*[p.encode() for p in outs])
    status = (C.c_int32 * len(names))()
    rc = sg_lib.sg_pack_build_many(a_src, a_names, a_out, len(names), threads, status)
    return rc, list(status), outs


def test_native_pack_builder_writes_what_the_python_builder_writes(sg_lib, tmp_path, golden_index):
    """sg_pack_build_many (zip + pickle + seg.json + staging + pack write in C++ threads) against cache.pack_scene (zipfile, pickle, NumPy):
    the same bytes, for V == N, V != N with a non-identity unmapper, and a scene with one-point segments."""
    from seggroup_amd import cache
    root, names = _tree(tmp_path, golden_index)
    rc, status, outs = _native_build(sg_lib, root, names)
    assert rc == len(names) and status == [0] * len(names), (rc, status, sg_lib.sg_last_error())
    for n, out in zip(names, outs):
        want = open(cache.pack_scene(root, n, force=True), "rb").read()
        assert open(out, "rb").read() == want, n
        assert not [f for f in os.listdir(os.path.dirname(out)) if ".tmp" in f]          # written atomically, nothing left behind


def test_native_pack_builder_refuses_what_it_does_not_know_and_the_python_builder_takes_over(sg_lib, tmp_path, golden_index, monkeypatch):
    import numpy as np
    import torch
    from seggroup_amd import cache
    root, names = _tree(tmp_path, golden_index, names=("tiny_4k",))
    src = cache.source_files(root, names[0])
    good = {p: open(p, "rb").read() for p in src}

    def status_with(path, data):
        open(path, "wb").write(data)
        rc, status, _ = _native_build(sg_lib, root, names[:1], threads=1)
        msg = sg_lib.sg_last_error().decode()
        open(path, "wb").write(good[path])
        return rc, status[0], msg
    # a truncated archive, a file that is not a zip, a tensor view that reaches beyond its storage, a compressed archive, a dict instead of a tensor
    rc, st, msg = status_with(src[0], good[src[0]][:len(good[src[0]]) // 2])
    assert rc == 0 and st < 0 and "sg_pack_build" in msg
    rc, st, msg = status_with(src[5], b"not a zip archive at all" * 10)
    assert rc == 0 and st < 0
    tmp = os.path.join(str(tmp_path), "t.pth")
    import zipfile
    with zipfile.ZipFile(tmp + ".z", "w", zipfile.ZIP_DEFLATED) as zf:   # ... a compressed archive is not
        with zipfile.ZipFile(src[1]) as zin:
            for n in zin.namelist():
                zf.writestr(n, zin.read(n))
    rc, st, msg = status_with(src[1], open(tmp + ".z", "rb").read())
    assert rc == 0 and st < 0 and ("compressed" in msg or "data.pkl" in msg)
    torch.save({"a": torch.zeros(3)}, tmp)
    rc, st, msg = status_with(src[2], open(tmp, "rb").read())
    assert rc == 0 and st < 0 and ("plain tensor file" in msg or "single tensor" in msg)
    # a pickle whose tensor claims 200 elements of a 4-element storage (the size operand of a real file patched): refused, not read out of bounds
    torch.save(torch.arange(4), tmp)
    with zipfile.ZipFile(tmp) as zin:
        members = {n: zin.read(n) for n in zin.namelist()}
    pk = [n for n in members if n.endswith("data.pkl")][0]
    at = members[pk].index(b"K\x04\x85")                                # BININT1 4, TUPLE1 = size (4,)
    members[pk] = members[pk][:at] + b"K\xc8\x85" + members[pk][at + 3:]
    with zipfile.ZipFile(tmp + ".big", "w", zipfile.ZIP_STORED) as zf:
        for n, v in members.items():
            zf.writestr(n, v)
    rc, st, msg = status_with(src[1], open(tmp + ".big", "rb").read())
    assert rc == 0 and st < 0 and "beyond its storage" in msg, msg
    # cache.build_missing: a scene the native builder refuses is built by the Python builder (here: a float64 point cloud is fine natively,
    # a scene NAME that needs JSON escaping is not)
    os.makedirs(os.path.join(root, "x"), exist_ok=True)
    calls = []
    real = cache.pack_scene
    monkeypatch.setattr(cache, "pack_scene", lambda *a, **k: (calls.append(a[1]), real(*a, **k))[1])
    monkeypatch.setattr(cache, "_build_native", lambda root_, todo, style, workers: list(todo))       # "refused everything"
    assert cache.build_missing(root, names, workers=2) == len(names) and sorted(calls) == sorted(names)
    assert all(cache.is_current(root, n) for n in names)


# ---- pack loader (csrc/loader.cpp), host side: SG_LOADER_DRY=1 makes no HIP call (tools/host_scale_rehearsal.py's mode) --------------
def test_pack_loader_host_side_without_a_gpu(sg_lib, tmp_path, golden_index, monkeypatch):
    """The loader's host side without a GPU: every ticket completes -- more packs than slots, a file that is no pack in the middle -- and every
    scene carries its own header's sizes, its own per-segment arrays and seg_of_vertex table, and an adjacency address behind the file's bytes
    (the int32 rows are widened to the kernels' int64 in the worker's staging buffer, csrc/loader.cpp)."""
    import ctypes as C
    import struct
    from seggroup_amd import cache, hip
    from seggroup_amd.scene import seg_of_vertex
    if not hasattr(sg_lib, "sg_loader_create_sized"):
        pytest.skip("the sanitizer build holds the host-only sources; the loader is compiled with hipcc")
    monkeypatch.setenv("SG_LOADER_DRY", "1")
    root, names = _tree(tmp_path, golden_index)
    packs = [cache.pack_scene(root, n, force=True) for n in names]
    bad = str(tmp_path / "not_a_pack.sgpack")
    open(bad, "wb").write(b"SGPACK00" + b"\0" * 64)
    order = [packs[i % len(packs)] for i in range(17)]
    order.insert(7, bad)
    L = sg_lib.sg_loader_create_sized(3, 4, max(os.path.getsize(p) for p in packs), max(cache.pack_dims(p)["E0"] for p in packs))
    assert L, sg_lib.sg_last_error()
    try:
        tickets = []
        for p in order[:6]:
            t = sg_lib.sg_loader_submit(L, p.encode()); assert t > 0; tickets.append((t, p))
        nxt, seen = 6, 0
        while tickets:
            t, p = tickets.pop(0)
            sc, slot, name = hip.Scene(), C.c_int(-1), C.create_string_buffer(256)
            rc = sg_lib.sg_loader_wait(L, t, C.byref(sc), C.byref(slot), name, 256)
            if p == bad:
                assert rc < 0 and b"not a scene pack" in sg_lib.sg_last_error()
            else:
                assert rc == 0, sg_lib.sg_last_error()
                d = cache.pack_dims(p)
                assert (sc.N, sc.S, sc.E0, sc.V) == (d["N"], d["S"], d["E0"], d["V"]) and name.value.decode() in p
                with open(p, "rb") as f:
                    f.seek(8); (hlen,) = struct.unpack("<I", f.read(4)); hdr = json.loads(f.read(hlen)); body = f.read()
                def arr(k):
                    dt, shape, off = hdr["arrays"][k]
                    return np.frombuffer(body, dtype=np.dtype(dt), count=int(np.prod(shape)), offset=off).reshape(shape)
                got = np.ctypeslib.as_array(C.cast(sc.h_seg_of_vertex, C.POINTER(C.c_int32)), shape=(sc.V,))
                assert np.array_equal(got, seg_of_vertex(arr("seg_of_point"), arr("unmap")))
                for k, ptr in (("seg_first", sc.h_seg_first), ("seg_size", sc.h_seg_size), ("seg_ins", sc.h_seg_ins), ("seg_sem", sc.h_seg_sem)):
                    assert np.array_equal(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(sc.S,)), arr(k)), k
                # dry: device addresses are offsets into a slot that starts at null; the widened adjacency lies behind the file's bytes
                if arr("adj").dtype == np.int32:
                    assert (sc.d_adj or 0) >= len(body) and (sc.d_adj or 0) % 256 == 0
                assert sg_lib.sg_loader_release(L, slot.value) == 0
                seen += 1
            if nxt < len(order):
                t2 = sg_lib.sg_loader_submit(L, order[nxt].encode()); assert t2 > 0; tickets.append((t2, order[nxt])); nxt += 1
        assert seen == 17
    finally:
        sg_lib.sg_loader_destroy(L)
