"""One process per GPU: keep a rank's host side on the NUMA node its GPU hangs off.

A rank's engine-group threads build descriptors and read results out of pinned memory, its writer workers format megabytes of
label text per scene, its loader threads fill pinned staging buffers -- at the headline rate that is tens of GB/s of host
memory traffic per GPU.  On an 8-GPU node with the ranks' threads scheduled anywhere, half of that crosses the socket link.
`bind_to_gpu_node(device)` restricts the calling process -- EVERY thread it has at that moment (`/proc/self/task`: asking for the
device's PCI address has already started the HIP runtime's threads; `sched_setaffinity(0, ...)` alone would move only the caller) and
so every thread started afterwards, native ones included -- to the CPUs of the GPU's NUMA node; first-touch then places its pinned
buffers there too.  Nothing here needs root, and nothing changes on a single-node box (numa_node = -1, or one node): the function
reports what it found and returns.  `unbound()` is the opt-out for a leg that wants the whole host back (bench.py's CPU baseline).
"""
from __future__ import annotations

import os
from typing import Dict, Optional


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text: str):
    cpus = set()
    for part in text.split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            a, b = part.split('-')
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_numa_node(device_index: int) -> Dict[str, object]:
    """PCI address and NUMA node of a HIP device (sysfs; -1 = the platform reports none)."""
    import torch
    props = torch.cuda.get_device_properties(device_index)
    bdf = None
    if all(hasattr(props, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    node = -1
    if bdf is not None:
        txt = _read(f"/sys/bus/pci/devices/{bdf}/numa_node")
        if txt is not None:
            try:
                node = int(txt)
            except ValueError:
                node = -1
    return {"pci": bdf, "numa_node": node}


def set_affinity_all_threads(cpus) -> int:
    """sched_setaffinity for every thread this process has right now; returns how many were moved (a thread that exits in between is
    skipped).  Threads created later inherit their creator's mask."""
    moved = 0
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
            moved += 1
        except OSError:
            pass
    return moved


_unbound_mask = None                                  # the affinity mask the process had before the first bind


class unbound:
    """Context manager: the process's threads get the pre-bind CPU mask back for the duration of a leg, then the bound one again."""

    def __enter__(self):
        self.prev = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
        if self.prev is not None and _unbound_mask is not None and _unbound_mask != self.prev:
            set_affinity_all_threads(_unbound_mask)
        return self

    def __exit__(self, *exc):
        if self.prev is not None and _unbound_mask is not None and _unbound_mask != self.prev:
            set_affinity_all_threads(self.prev)
        return False


def bind_to_gpu_node(device_index: int, mode: str = "auto", node_override: Optional[int] = None) -> Dict[str, object]:
    """Restrict this process (all of its current threads) to the CPUs of `device_index`'s NUMA node (mode 'auto'); 'off' only reports.
    `node_override` (or env SG_NUMA_NODE) names the node instead of sysfs -- tools/host_scale_rehearsal.py spreads eight pretend ranks
    over a host's sockets with it.  Returns what was found and done: {'pci', 'numa_node', 'cpus_before', 'cpus_after', 'bound', 'threads'}."""
    global _unbound_mask
    info = gpu_numa_node(device_index)
    if node_override is None and os.environ.get("SG_NUMA_NODE", "") != "":
        node_override = int(os.environ["SG_NUMA_NODE"])
    if node_override is not None:
        info["numa_node"], info["node_override"] = int(node_override), True
    before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set()
    if _unbound_mask is None:
        _unbound_mask = set(before)
    info.update(cpus_before=len(before), cpus_after=len(before), bound=False)
    if mode == "off" or info["numa_node"] < 0 or not before:
        return info
    nodes = [d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()] if os.path.isdir("/sys/devices/system/node") else []
    if len(nodes) < 2:
        return info
    txt = _read(f"/sys/devices/system/node/node{info['numa_node']}/cpulist")
    if not txt:
        return info
    want = parse_cpulist(txt) & before            # never leave the cpuset the container grants
    if not want:
        return info
    moved = set_affinity_all_threads(want)
    info.update(cpus_after=len(want), bound=True, threads=moved)
    return info
