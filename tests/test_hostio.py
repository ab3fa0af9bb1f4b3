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
