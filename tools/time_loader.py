#!/usr/bin/env python3
"""The native pack loader alone, and loader + engine without file output: where does the driver's rate go?
    python tools/time_loader.py [--scenes 1024] [--base /dev/shm]"""
import argparse, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--base", default="/dev/shm")
    ap.add_argument("--threads", default="4,8,16,32")
    a = ap.parse_args()
    import numpy as np, torch
    from seggroup_amd import cache, hip, synthetic, weights
    from seggroup_amd.model import BatchRunner
    root = tempfile.mkdtemp(prefix="sg_loader_", dir=a.base)
    try:
        paths = []
        for i in range(a.distinct):
            sc = synthetic.make_scene(150000, 1500, 20004 + i % 4, name=f"s{i:04d}")
            p = os.path.join(root, f"s{i:04d}.sgpack")
            cache.write_pack(p, sc.name, cache.stage_arrays(sc.data, sc.weak_label, sc.seg, sc.adj, sc.unmap, sc.gt))
            paths.append(p)
        for i in range(a.distinct, a.scenes):
            p = os.path.join(root, f"s{i:04d}.sgpack")
            os.symlink(paths[i % a.distinct], p)
            paths.append(p)
        size = os.path.getsize(paths[0])
        torch.cuda.set_device(0)
        for nt in [int(x) for x in a.threads.split(",")]:
            ld = cache.PackLoader(threads=nt, slots=192, slot_bytes=size)
            t0 = time.perf_counter()
            tk = [ld.submit(p) for p in paths[:192]]
            nxt = 192
            done = 0
            while tk:
                s = ld.wait(tk.pop(0))
                s.release()
                done += 1
                if nxt < len(paths):
                    tk.append(ld.submit(paths[nxt])); nxt += 1
            dt = time.perf_counter() - t0
            print(f"loader alone, {nt:3d} threads: {done / dt:8.1f} scenes/s ({size / 1e6:.1f} MB packs: {done * size / dt / 1e9:.1f} GB/s)", flush=True)
            ld.close()
        # loader + engine, no files
        w = weights.make_weights(1, 2.0)
        for nt in (8, 16):
            for transfer in ("tables", "full"):
                ld = cache.PackLoader(threads=nt, slots=192, slot_bytes=size)
                batches = [paths[k:k + 64] for k in range(0, len(paths), 64)]
                pend = [ld.submit(p) for p in batches[0]]
                runner, tickets = None, []
                t0 = time.perf_counter()
                n = 0
                def consume(t):
                    nonlocal n
                    for s_, r in zip(t.scenes, runner.wait(t)):
                        n += 1
                        s_.release()
                for bi in range(len(batches)):
                    scenes = [ld.wait(t) for t in pend]
                    pend = [ld.submit(p) for p in batches[bi + 1]] if bi + 1 < len(batches) else []
                    if runner is None:
                        runner = BatchRunner(w, scenes, inflight=80, label_transfer=transfer)
                        t0 = time.perf_counter(); n0 = n
                    tickets.append(runner.submit(scenes, hip.MODE_INS_INFER))
                    if len(tickets) > 1:
                        consume(tickets.pop(0))
                while tickets:
                    consume(tickets.pop(0))
                dt = time.perf_counter() - t0
                print(f"loader ({nt} threads) + engine, label transfer {transfer}: {n / dt:8.1f} scenes/s", flush=True)
                runner.close(); ld.close()
    finally:
        shutil.rmtree(root, ignore_errors=True)


main()
