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
