#!/usr/bin/env python3
"""Does bulk H2D traffic (what the pack loader generates: ~13 MB per scene) slow the scene engine down, with the scenes themselves resident?
Engine on 64 resident scenes at full rate while N background threads copy 13 MB pinned buffers to the device back to back.
    python tools/exp_h2d_interference.py [--mb 13] [--threads 0,2,4,8]"""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=13.0)
    ap.add_argument("--threads", default="0,2,4,8")
    ap.add_argument("--chunk-mb", type=float, default=0.0, help="split every copy into chunks of this size (0 = one copy)")
    ap.add_argument("--scene-cache", default=os.environ.get("SG_SCENE_CACHE", ""))
    a = ap.parse_args()
    jobs = [(150000, 1500, 30000 + i, "voronoi", a.scene_cache) for i in range(64)]
    it, pool = bench.generate_scenes(jobs, 16)
    import torch
    from seggroup_amd import hip, weights
    from seggroup_amd.model import BatchRunner
    from seggroup_amd.scene import DeviceScene
    scenes = [DeviceScene.from_synthetic(s, device="cuda:0") for s in it]
    if pool is not None:
        pool.shutdown()
    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
    runner = BatchRunner(W, scenes, inflight=80, device="cuda:0", timing=0)
    nbytes = int(a.mb * 1e6)
    for nt in [int(x) for x in a.threads.split(",")]:
        stop = threading.Event()
        copied = [0] * max(nt, 1)

        def copier(k):
            torch.cuda.set_device(0)
            src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
            st = torch.cuda.Stream()
            ch = int(a.chunk_mb * 1e6)
            with torch.cuda.stream(st):
                while not stop.is_set():
                    if ch > 0:
                        for o in range(0, nbytes, ch):
                            dst[o:o + ch].copy_(src[o:o + ch], non_blocking=True)
                    else:
                        dst.copy_(src, non_blocking=True)
                    st.synchronize()
                    copied[k] += 1
        ths = [threading.Thread(target=copier, args=(k,)) for k in range(nt)]
        for t in ths:
            t.start()
        for _ in range(3):
            runner.run(scenes, hip.MODE_INS_INFER)
        c0 = sum(copied)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pend, steps = [], 40
        for _ in range(steps):
            pend.append(runner.submit(scenes, hip.MODE_INS_INFER))
            if len(pend) > 2:
                runner.wait(pend.pop(0))
        while pend:
            runner.wait(pend.pop(0))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c1 = sum(copied)
        stop.set()
        for t in ths:
            t.join()
        print(f"{nt} copy threads: engine {steps * 64 / dt:8.1f} scenes/s | background H2D {(c1 - c0) / dt:8.1f} copies/s of {a.mb} MB = {(c1 - c0) * nbytes / dt / 1e9:5.1f} GB/s", flush=True)
    runner.close()


if __name__ == "__main__":
    main()
