"""The sorting / merging networks of csrc/knn_device.h (list_merge12), checked on the CPU from the header's own text: the 12-key sorter
with the 0-1 principle (all 4096 inputs), the pruned bitonic merge against `sorted(list + batch)[:20]` on random and partly empty inputs.
The device runs the same exchanges on doubles (`sg_selftest_list_insert`, -m gpu); this test pins the index tables."""
import os
import random
import re

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seggroup_amd", "csrc", "knn_device.h")


def _body():
    src = open(HEADER).read()
    return src[src.index("void list_merge12"):src.index("// upper bound of the score")]


def _sorter():
    return [(int(m.group(1)), int(m.group(2))) for m in re.finditer(r"list_cx\(b\[(\d+)\], b\[(\d+)\]\);", _body())]


def test_twelve_key_sorter_sorts_every_01_input():
    net = _sorter()
    assert len(net) == 39
    for bits in range(1 << 12):
        v = [(bits >> i) & 1 for i in range(12)]
        for a, b in net:
            if v[a] < v[b]:
                v[a], v[b] = v[b], v[a]
        assert all(v[i] >= v[i + 1] for i in range(11)), bits


def _loops(body):
    """The merge part as (kind, dst array, dst index expr, src array, src index expr, range) tuples, read from the header text."""
    out = []
    pat = re.compile(r"for \(int i = 0; i < (\d+); \+\+i\) (list_cx\((\w+)\[([^\]]+)\], (\w+)\[([^\]]+)\]\)|(\w+)\[([^\]]+)\] = list_max\((\w+)\[([^\]]+)\], (\w+)\[([^\]]+)\]\));")
    for m in pat.finditer(body):
        n = int(m.group(1))
        if m.group(2).startswith("list_cx"):
            out.append(("cx", m.group(3), m.group(4), m.group(5), m.group(6), n))
        else:
            assert (m.group(7), m.group(8)) == (m.group(9), m.group(10))
            out.append(("max", m.group(7), m.group(8), m.group(11), m.group(12), n))
    return out


def _merge(kv, b):
    body = _body()
    kv, b = list(kv), list(b)
    arr = {"kv": kv, "b": b}

    def cx(x, i, y, j):
        hi, lo = max(x[i], y[j]), min(x[i], y[j])
        x[i], y[j] = hi, lo

    for a_, b_ in _sorter():
        cx(b, a_, b, b_)
    loops = _loops(body)
    assert [l[0] for l in loops] == ["cx", "max", "max", "max"]
    for kind, da, di, sa, si, n in loops:
        for i in range(n):
            d, s = eval(di, {"i": i}), eval(si, {"i": i})
            if kind == "cx":
                cx(arr[da], d, arr[sa], s)
            else:
                arr[da][d] = max(arr[da][d], arr[sa][s])
    tail = re.findall(r"list_cx\(kv\[(\d+)\], kv\[(\d+)\]\);", body)
    assert len(tail) == 4
    for a_, b_ in tail:
        cx(kv, int(a_), kv, int(b_))
    assert "for (int d = 8; d > 0; d >>= 1)" in body and "if ((i & d) == 0) list_cx(kv[i], kv[i + d]);" in body
    d = 8
    while d:
        for i in range(16):
            if not i & d:
                cx(kv, i, kv, i + d)
        d >>= 1
    return kv


def test_merge_of_twelve_leaves_the_top_twenty_of_the_union():
    rnd = random.Random(5)
    for _ in range(4000):
        vals = rnd.sample(range(1, 10 ** 6), 32)
        k = rnd.randint(0, 20)
        kv = sorted(vals[:k], reverse=True) + [0] * (20 - k)
        k2 = rnd.randint(0, 12)
        b = vals[20:20 + k2] + [0] * (12 - k2)
        rnd.shuffle(b)
        assert _merge(kv, b) == sorted(kv + b, reverse=True)[:20]
