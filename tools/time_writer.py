#!/usr/bin/env python3
"""The label-file writer pool alone (sg_writer_*, capi.cpp): scenes per second it can put on the file system from a host buffer, by thread
count and format -- the ceiling of bench.py's with_label_files_scenes_per_s leg.  No GPU involved.

    python tools/time_writer.py [dir] [V] [scenes]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seggroup_amd import hip  # noqa: E402

base = sys.argv[1] if len(sys.argv) > 1 else tempfile.gettempdir()
V = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
lib = hip.lib()
labels = np.random.default_rng(0).integers(-1, 40, (14, V)).astype(np.int32)
print("file system of %s: %s" % (base, subprocess.run(["df", "-T", base], capture_output=True, text=True).stdout.splitlines()[-1]))
print("cpus: %d" % os.cpu_count())
for fm, name in ((2, "npy"), (3, "txt+npy")):
    for threads in (4, 8, 16, 32):
        with tempfile.TemporaryDirectory(prefix="sgw_", dir=base) as td:
            dirs = [os.path.join(td, "scene%04d" % i) for i in range(64)]
            for d in dirs:
                os.makedirs(d)
            w = lib.sg_writer_create(threads, 256)
            best = 0.0
            for rep in range(3):                       # pass 0 creates the files, later passes overwrite them (like the bench leg)
                t = time.perf_counter()
                for i in range(n):
                    hip.check(lib.sg_writer_submit_scene(w, dirs[i % 64].encode(), labels.ctypes.data, V, 14, fm, rep * n + i))
                hip.check(lib.sg_writer_flush(w))
                dt = time.perf_counter() - t
                best = max(best, n / dt)
                if rep == 0:
                    first = n / dt
            lib.sg_writer_destroy(w)
        print("%-8s %2d threads: %7.0f scenes/s creating, %7.0f overwriting (%.1f GB/s of labels)" % (name, threads, first, best, best * 14 * V * 4 / 1e9))
