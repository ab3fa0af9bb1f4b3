#!/usr/bin/env python3
"""A one-off parity sweep on FRESH seeds (round 6, after the kNN's list-form thresholds and the split bucket step): scenes no fixture, scan or test has
seen -- uniform and ScanNet-shaped at 150k points, 60k / 600 ScanNet-shaped, 3,000 / 30 and 3,000 / 150 with one-point segments -- through the engine
(10 x 8) and through the oracle (oracle/cpu_ref.py, a pool of worker processes started before this process touches the GPU): all 14 label vectors, the
cluster trace and the two integer metric tensors must be equal.  Writes profiles/<tag>_parity_sweep.json.

    python3 tools/parity_sweep.py [--tag r06] [--full 16] [--workers 24]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def spec(full, off=0):
    out = [(150000, 1500, 62000 + off + i, {}) for i in range(full)]
    out += [(150000, 1500, 72000 + off + i, {"seg_profile": "scannet"}) for i in range(max(2, full // 2))]
    out += [(60000, 600, 72100 + off + i, {"seg_profile": "scannet"}) for i in range(4)]
    out += [(3000, 30, 62100 + off + i, {}) for i in range(8)] + [(3000, 150, 62200 + off + i, {"min_seg": 1}) for i in range(8)]
    out += [(20000, 200, 62300 + off + i, {"dup_frac": 0.1}) for i in range(4)]
    return out


def digest(labels, trace, iou_sem, iou_ins):
    h = hashlib.sha256()
    for v in labels:
        h.update(np.ascontiguousarray(v, dtype=np.int32).tobytes())
    h.update(np.asarray(trace, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(iou_sem).tobytes())
    h.update(np.ascontiguousarray(iou_ins).tobytes())
    return h.hexdigest()


def oracle_job(job):
    n, s, seed, kw, mode = job
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    from threadpoolctl import threadpool_limits
    from oracle import cpu_ref
    from seggroup_amd import hip, synthetic, weights
    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
    sc = synthetic.make_scene(n, s, seed, **kw)
    t = time.time()
    with threadpool_limits(limits=4):
        ref = cpu_ref.forward_scene(sc, W, mode)
    names = [f"layer_{l}.{k}" for l in (1, 2, 3, 4) for k in ("seg", "ins", "sem")] + ["final.ins", "final.sem"]
    return sc, digest([ref["labels"][nm] for nm in names], ref["trace"], ref["metrics"][0], ref["metrics"][1]), list(ref["trace"]), time.time() - t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r06")
    ap.add_argument("--full", type=int, default=16)
    ap.add_argument("--workers", type=int, default=24)
    ap.add_argument("--seed-offset", type=int, default=0, help="added to every seed of the list: a second sweep sees other scenes")
    ap.add_argument("--groups", type=int, default=14)
    ap.add_argument("--suffix", default="", help="profiles/<tag>_parity_sweep<suffix>.json")
    a = ap.parse_args()
    jobs = spec(a.full, a.seed_offset)
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(a.workers, len(jobs)), mp_context=mp.get_context("spawn")) as pool:
        refs = list(pool.map(oracle_job, [j + ("ins_infer",) for j in jobs]))
    t_oracle = time.time() - t0
    from seggroup_amd import hip, weights
    from seggroup_amd.model import Engine
    from seggroup_amd.scene import DeviceScene
    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
    mode = hip.MODE_INS_INFER
    scenes = [DeviceScene.from_synthetic(r[0], device="cuda:0") for r in refs]
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    eng = Engine(W, caps, groups=a.groups, per_group=8, device="cuda:0", timing=0)
    rows, bad = [], 0
    for rep in range(2):
        res = eng.run(scenes, mode)
        for (n, s, seed, kw), r, (sc, want, trace, secs) in zip(jobs, res, refs):
            got = digest([r.labels[i] for i in range(r.n_vectors)], r.trace, r.iou_sem, r.iou_ins)
            ok = got == want
            bad += not ok
            if rep == 0:
                rows.append({"points": n, "segments": s, "seed": seed, "kw": kw, "trace": trace, "equal": ok, "oracle_s": round(secs, 1)})
            elif not ok:
                rows[len(rows) - len(jobs) + jobs.index((n, s, seed, kw))]["equal"] = False
    eng.close()
    out = {"what": f"fresh seeds through the engine ({a.groups} x 8, two runs, ins_infer, seed offset {a.seed_offset}) against oracle/cpu_ref.py: sha256 over the 14 label vectors + cluster trace + the integer metric tensors",
           "scenes": len(jobs), "unequal": bad, "oracle_pool_s": round(t_oracle, 1), "rows": rows}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{a.tag}_parity_sweep{a.suffix}.json"), "w"), indent=1)
    print(f"{len(jobs)} fresh scenes x 2 runs: {bad} results differ from the oracle; oracle pool {t_oracle:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
