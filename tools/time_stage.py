#!/usr/bin/env python3
"""The driver's staging step alone (seggroup_amd/cache.py: load_pack = read the scene pack into a pinned buffer + ONE upload + typed
views): scenes per second by loader-thread count, against the engine's rate the loaders have to feed.

    python tools/time_stage.py [--scenes 128] [--base /dev/shm]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=128)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--segments", type=int, default=1500)
    ap.add_argument("--base", default="/dev/shm")
    a = ap.parse_args()
    import torch
    from seggroup_amd import cache, synthetic

    root = tempfile.mkdtemp(prefix="sg_stage_", dir=a.base)
    try:
        base = [synthetic.make_scene(a.points, a.segments, 20004 + i, name=f"scene{i:04d}_00") for i in range(4)]
        scenes = []
        for i in range(a.scenes):
            b = base[i % 4]
            scenes.append(synthetic.Scene(f"scene{i:04d}_00", b.data, b.weak_label, b.seg, b.adj, b.unmap, b.gt))
        synthetic.write_reference_tree(root, scenes)
        names = [s.name for s in scenes]
        t = time.time()
        cache.build_missing(root, names, "manual", workers=16)
        print("packs built in %.1f s; one pack = %.1f MB" % (time.time() - t, os.path.getsize(cache.pack_scene(root, names[0], "manual")) / 1e6))
        paths = [cache.pack_scene(root, n, "manual") for n in names]
        cache.load_pack(paths[0], device="cuda:0")
        for threads in (1, 4, 8, 16, 32, 64):
            pool = ThreadPoolExecutor(max_workers=threads)
            list(pool.map(lambda p: cache.load_pack(p, device="cuda:0"), paths[:threads]))       # per-thread pinned buffers and streams
            torch.cuda.synchronize()
            t = time.time()
            keep = list(pool.map(lambda p: cache.load_pack(p, device="cuda:0"), paths))
            torch.cuda.synchronize()
            dt = time.time() - t
            del keep
            pool.shutdown()
            print("%2d loader threads: %6.0f scenes/s (%.2f ms per scene and thread)" % (threads, len(paths) / dt, dt * threads / len(paths) * 1e3))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
