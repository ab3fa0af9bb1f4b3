"""seggroup_amd.numa: CPU-list parsing and the no-op behaviour on a box without NUMA information (no GPU needed for these)."""
import os

from seggroup_amd import numa


def test_parse_cpulist():
    assert numa.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    assert numa.parse_cpulist("") == set()
    assert numa.parse_cpulist("5") == {5}


def test_bind_is_a_no_op_without_numa_information(monkeypatch):
    monkeypatch.setattr(numa, "gpu_numa_node", lambda i: {"pci": None, "numa_node": -1})
    before = os.sched_getaffinity(0)
    info = numa.bind_to_gpu_node(0, "auto")
    assert not info["bound"] and os.sched_getaffinity(0) == before and info["cpus_after"] == len(before)


def test_bind_restricts_to_the_nodes_cpus_inside_the_granted_set(monkeypatch, tmp_path):
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        return
    keep = sorted(before)[:max(1, len(before) // 2)]
    monkeypatch.setattr(numa, "gpu_numa_node", lambda i: {"pci": "0000:00:00.0", "numa_node": 1})
    real_read, real_listdir, real_isdir = numa._read, os.listdir, os.path.isdir
    monkeypatch.setattr(numa, "_read", lambda p: ",".join(map(str, keep + [10 ** 6])) if p.endswith("node1/cpulist") else real_read(p))
    monkeypatch.setattr(os, "listdir", lambda p: ["node0", "node1"] if p == "/sys/devices/system/node" else real_listdir(p))
    monkeypatch.setattr(os.path, "isdir", lambda p: True if p == "/sys/devices/system/node" else real_isdir(p))
    try:
        info = numa.bind_to_gpu_node(0, "auto")
        assert info["bound"] and os.sched_getaffinity(0) == set(keep) and info["cpus_after"] == len(keep)
        assert not numa.bind_to_gpu_node(0, "off")["bound"]
    finally:
        os.sched_setaffinity(0, before)
