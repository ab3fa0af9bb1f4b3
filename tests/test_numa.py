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


def test_bind_moves_threads_that_already_exist_and_unbound_gives_the_host_back(monkeypatch):
    """sched_setaffinity(0, ...) binds only the caller; the HIP runtime's threads exist before the bind."""
    import threading
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        return
    keep = sorted(before)[:max(1, len(before) // 2)]
    go, stop, looked, seen = threading.Event(), threading.Event(), threading.Event(), {}

    def early():                                             # a thread that exists before the bind
        go.wait(10)
        seen["bound"] = os.sched_getaffinity(0)
        looked.set()
        stop.wait(10)
        seen["restored"] = os.sched_getaffinity(0)
    th = threading.Thread(target=early)
    th.start()
    monkeypatch.setattr(numa, "gpu_numa_node", lambda i: {"pci": "0000:00:00.0", "numa_node": 1})
    monkeypatch.setattr(numa, "_unbound_mask", None)
    real_read, real_listdir, real_isdir = numa._read, os.listdir, os.path.isdir
    monkeypatch.setattr(numa, "_read", lambda p: ",".join(map(str, keep)) if p.endswith("node1/cpulist") else real_read(p))
    monkeypatch.setattr(os, "listdir", lambda p: ["node0", "node1"] if p == "/sys/devices/system/node" else real_listdir(p))
    monkeypatch.setattr(os.path, "isdir", lambda p: True if p == "/sys/devices/system/node" else real_isdir(p))
    try:
        info = numa.bind_to_gpu_node(0, "auto")
        assert info["bound"] and info["threads"] >= 2
        go.set()
        assert looked.wait(10)                               # the early thread has read its mask before the leg below changes it again
        with numa.unbound():
            assert os.sched_getaffinity(0) == before         # the CPU-baseline leg sees the whole host
        assert os.sched_getaffinity(0) == set(keep)
        numa.set_affinity_all_threads(before)
        stop.set()
        th.join()
        assert seen["bound"] == set(keep) and seen["restored"] == before
    finally:
        go.set(); stop.set()
        numa.set_affinity_all_threads(before)


def test_node_override_names_the_node(monkeypatch):
    monkeypatch.setattr(numa, "gpu_numa_node", lambda i: {"pci": None, "numa_node": -1})
    info = numa.bind_to_gpu_node(0, "off", node_override=3)
    assert info["numa_node"] == 3 and info["node_override"] and not info["bound"]


def test_eight_gpus_on_two_nodes_bind_each_rank_to_its_gpus_node(monkeypatch):
    """An 8-GPU MI355X node as the platform describes it: GPUs 0-3 behind socket 0, 4-7 behind socket 1 (PCI address -> numa_node in sysfs).
    Every rank r binds to the CPUs of GPU r's node, inside whatever cpuset the container grants; here the map is faked over this host's CPUs."""
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        return
    cpus = sorted(before)
    half = len(cpus) // 2
    node_cpus = {0: cpus[:half], 1: cpus[half:]}
    pci = {d: "0000:%02x:00.0" % (0x05 + 0x10 * d) for d in range(8)}
    monkeypatch.setattr(numa, "gpu_numa_node", lambda i: {"pci": pci[i], "numa_node": 0 if i < 4 else 1})
    real_read, real_listdir, real_isdir = numa._read, os.listdir, os.path.isdir

    def fake_read(p):
        for n, cl in node_cpus.items():
            if p.endswith(f"node{n}/cpulist"):
                return ",".join(map(str, cl))
        return real_read(p)
    monkeypatch.setattr(numa, "_read", fake_read)
    monkeypatch.setattr(os, "listdir", lambda p: ["node0", "node1"] if p == "/sys/devices/system/node" else real_listdir(p))
    monkeypatch.setattr(os.path, "isdir", lambda p: True if p == "/sys/devices/system/node" else real_isdir(p))
    try:
        for dev in range(8):                                       # what rank `dev` of eight does at start-up (infer.run_worker, bench.py)
            numa.set_affinity_all_threads(before)
            monkeypatch.setattr(numa, "_unbound_mask", None)
            info = numa.bind_to_gpu_node(dev, "auto")
            assert info["bound"] and info["pci"] == pci[dev] and info["numa_node"] == (0 if dev < 4 else 1)
            assert os.sched_getaffinity(0) == set(node_cpus[info["numa_node"]])
            assert info["cpus_before"] == len(before) and info["cpus_after"] == len(node_cpus[info["numa_node"]])
            with numa.unbound():
                assert os.sched_getaffinity(0) == before
    finally:
        numa.set_affinity_all_threads(before)
