#!/usr/bin/env python3
"""Host-side rehearsal of W = 1, 2, 4, 8 ranks on ONE box (reference infer.py:93-101,232-237: one process per GPU,
workers / ngpus loader processes each).  No rank touches a GPU: what is rehearsed is everything a rank does on the HOST at the rate its GPU
would demand -- the native pack loader (file reads from tmpfs, header parsing, seg_of_vertex; SG_LOADER_DRY=1 replaces the upload by one pass
of reads over the staging buffer, which is what the copy engine's DMA does to host memory), the native writer pool (label tables expanded and
formatted by its workers, files written), NUMA binding as rank r of 8 would get it (ranks spread over the host's nodes) -- with the GPU stage
replaced by a pacing sleep of batch / RATE seconds.  A second leg emulates the bench's FULL label transfer: every scene's 8.4 MB of label
vectors land in a host buffer (a memcpy standing in for the DMA write at 3,000 x 8.4 MB = 25 GB/s per rank) and are written by reference.

    python tools/host_scale_rehearsal.py [--ranks 1,2,4,8] [--scenes 768] [--rate 3000] [--base /dev/shm] [--out profiles/r05_host_scale.json]

Reports, per output format and W: aggregate scenes/s, the slowest rank, and the ratio to W x the single-rank figure."""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE = [1500, 1006, 134, 87, 60]          # clusters per layer of a typical 150k / 1.5k scene: the value ranges of the label tables


def _rank(rank, world, cfg, barrier, q):
    os.environ["SEGGROUP_HOST_ONLY"] = "1"
    os.environ["SG_LOADER_DRY"] = "1"
    import numpy as np
    from seggroup_amd import hip, numa
    nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()) if os.path.isdir("/sys/devices/system/node") else [0]
    node = nodes[(rank * len(nodes)) // 8 % len(nodes)]                  # rank r of EIGHT: GPUs 0-3 on the first half of the nodes, 4-7 on the second
    numa.gpu_numa_node = lambda i: {"pci": "rehearsal", "numa_node": node}
    bound = numa.bind_to_gpu_node(0, cfg["numa"])
    lib = hip.lib()
    fm = (1 if "txt" in cfg["formats"] else 0) | (2 if "npy" in cfg["formats"] else 0)
    paths = cfg["paths"]
    mine = [paths[i % len(paths)] for i in range(rank, rank + cfg["scenes"] * world, world)]
    slot_bytes = max(os.path.getsize(p) for p in set(mine))
    L = lib.sg_loader_create_sized(cfg["loader_threads"], 4 * cfg["batch"], slot_bytes, 0)
    Wr = lib.sg_writer_create(cfg["writer_threads"], 256)
    assert L and Wr, lib.sg_last_error().decode()
    S = 1500
    rng = np.random.default_rng(rank)
    tables = np.empty((14, S), np.int32)
    for t in range(14):
        layer = min(t // 3, 4)
        kind = t % 3 if t < 12 else t - 11                                 # seg / ins / sem rows, like LABEL_NAMES
        tables[t] = rng.integers(0, TRACE[layer], S) if kind == 0 else (rng.integers(-1, 60, S) if kind == 1 else rng.integers(-1, 40, S))
    out_root = os.path.join(cfg["results_base"], "results", f"w{world}_r{rank}")
    dirs = [os.path.join(out_root, f"scene{i:04d}").encode() for i in range(cfg["out_dirs"])]
    for d in dirs:
        os.makedirs(d, exist_ok=True)
    full = cfg["full_labels"]
    ring, src = None, None
    B = cfg["batch"]
    batches = [mine[k:k + B] for k in range(0, len(mine), B)]
    barrier.wait()
    t0 = time.perf_counter()
    pend = [[lib.sg_loader_submit(L, p.encode()) for p in b] for b in batches[:2]]
    nxt, done, t_gpu_free, tag = 2, 0, t0, 0
    waits = {"loader": 0.0, "writer": 0.0, "pace": 0.0}
    only = cfg.get("only", "")
    kept = None                                                           # "writer only": the first batch's scenes, never released, reused
    for bi, b in enumerate(batches):
        ta = time.perf_counter()
        if only == "writer" and kept is not None:
            scenes, slots = kept, []
        else:
            scenes, slots = [], []
            for tk in pend.pop(0):
                sc, slot, name = hip.Scene(), C.c_int(-1), C.create_string_buffer(64)
                hip.check(lib.sg_loader_wait(L, tk, C.byref(sc), C.byref(slot), name, 64))
                scenes.append(sc); slots.append(slot.value)
            if only == "writer":
                kept, slots = scenes, []
                for extra in pend:                                        # drain what was requested ahead
                    for tk in extra:
                        sc, slot, name = hip.Scene(), C.c_int(-1), C.create_string_buffer(64)
                        hip.check(lib.sg_loader_wait(L, tk, C.byref(sc), C.byref(slot), name, 64))
                pend = []
        waits["loader"] += time.perf_counter() - ta
        if only != "writer" and nxt < len(batches):
            pend.append([lib.sg_loader_submit(L, p.encode()) for p in batches[nxt]]); nxt += 1
        # the GPU stage: this batch leaves the engine batch / RATE seconds after the engine was last free
        t_gpu_free = max(t_gpu_free, time.perf_counter()) + len(b) / cfg["rate"]
        ta = time.perf_counter()
        if t_gpu_free > ta:
            time.sleep(t_gpu_free - ta)
        waits["pace"] += time.perf_counter() - ta
        ta = time.perf_counter()
        for k, sc in enumerate(scenes if only != "loader" else []):
            tag += 1
            d = dirs[(done + k) % len(dirs)]
            if full:
                V = sc.V
                if ring is None:
                    src = np.zeros(14 * V, np.int32); src[:] = rng.integers(0, 1500, 14 * V, dtype=np.int32)
                    ring = [np.empty(14 * V, np.int32) for _ in range(3 * B)]
                buf = ring[tag % len(ring)]
                if tag > len(ring):
                    hip.check(lib.sg_writer_wait_tag(Wr, tag - len(ring)))     # the buffer's previous scene is on disk
                np.copyto(buf, src)                                           # stands in for the DMA write of the label vectors
                hip.check(lib.sg_writer_submit_scene(Wr, d, buf.ctypes.data, V, 14, fm, tag))
            else:
                hip.check(lib.sg_writer_submit_scene_tables(Wr, d, tables.ctypes.data, S, sc.h_seg_of_vertex, sc.V, 14, fm, tag))
        waits["writer"] += time.perf_counter() - ta
        for s_ in slots:
            hip.check(lib.sg_loader_release(L, s_))
        done += len(b)
    hip.check(lib.sg_writer_flush(Wr))
    dt = time.perf_counter() - t0
    lib.sg_writer_destroy(Wr)
    lib.sg_loader_destroy(L)
    q.put({"rank": rank, "scenes": done, "seconds": dt, "node": node, "bound": bool(bound.get("bound")), "cpus": bound.get("cpus_after"),
           "main_thread_s": {k: round(v, 3) for k, v in waits.items()}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--scenes", type=int, default=768, help="scenes per rank and leg")
    ap.add_argument("--rate", type=float, default=3000.0, help="scenes/s the pretend GPU of every rank sustains")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--base", default="/dev/shm")
    ap.add_argument("--numa", default="auto", choices=["auto", "off"])
    ap.add_argument("--results-base", default="", help="where the label files go (default: beside the packs, i.e. --base)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from seggroup_amd import cache, synthetic
    root = tempfile.mkdtemp(prefix="sg_rehearsal_", dir=a.base)
    report = {"what": "host-side rehearsal: W ranks on one box, GPU stage = pacing sleep, loader uploads = reads of the staging buffer",
              "rate_per_rank": a.rate, "scenes_per_rank": a.scenes, "host_cpus": os.cpu_count(), "legs": {}}
    try:
        paths = []
        for i in range(a.distinct):
            sc = synthetic.make_scene(150000, 1500, 20004 + i % 4, name=f"s{i:04d}")
            p = os.path.join(root, f"s{i:04d}.sgpack")
            cache.write_pack(p, sc.name, cache.stage_arrays(sc.data, sc.weak_label, sc.seg, sc.adj, sc.unmap, sc.gt))
            paths.append(p)
        ctx = mp.get_context("spawn")
        for leg, formats, full, lt, wt, only in (("npy, tables", "npy", False, 6, 6, ""), ("txt+npy, tables", "txt,npy", False, 8, 8, ""),
                                                 ("npy, full label vectors", "npy", True, 6, 6, ""), ("loader only (no files written)", "npy", False, 6, 6, "loader"),
                                                 ("writer only, npy, tables (one batch of scenes reused)", "npy", False, 6, 6, "writer")):
            rows = {}
            for W in [int(x) for x in a.ranks.split(",")]:
                cfg = {"paths": paths, "scenes": a.scenes, "rate": a.rate, "batch": a.batch, "formats": formats, "full_labels": full, "loader_threads": lt,
                       "writer_threads": wt, "root": root, "out_dirs": 96, "numa": a.numa, "only": only, "results_base": a.results_base or root}
                barrier, q = ctx.Barrier(W), ctx.Queue()
                procs = [ctx.Process(target=_rank, args=(r, W, cfg, barrier, q)) for r in range(W)]
                for p in procs:
                    p.start()
                res = [q.get(timeout=900) for _ in range(W)]
                for p in procs:
                    p.join(60)
                wall = max(r["seconds"] for r in res)
                rows[str(W)] = {"aggregate_scenes_per_s": round(sum(r["scenes"] for r in res) / wall, 1),
                                "slowest_rank_scenes_per_s": round(min(r["scenes"] / r["seconds"] for r in res), 1),
                                "nodes": sorted({r["node"] for r in res}), "cpus_per_rank": res[0]["cpus"],
                                "main_thread_s_rank0": next(r["main_thread_s"] for r in res if r["rank"] == 0)}
                shutil.rmtree(os.path.join(a.results_base or root, "results"), ignore_errors=True)
                print(leg, "W =", W, rows[str(W)], flush=True)
            base = rows.get("1", {}).get("aggregate_scenes_per_s")
            for W, r in rows.items():
                if base:
                    r["vs_W_times_single_rank"] = round(r["aggregate_scenes_per_s"] / (int(W) * base), 3)
            report["legs"][leg] = rows
        print(json.dumps(report))
        if a.out:
            os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
            json.dump(report, open(a.out, "w"), indent=1)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
