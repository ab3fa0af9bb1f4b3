#!/usr/bin/env python3
"""Device time of the pre-processing kernels on a ScanNet-sized synthetic scan, with the NumPy oracle
timed beside them on a bounded sample.   python tools/time_prepare.py [--out gpurun_out/prepare.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=600)
    ap.add_argument("--h", type=int, default=400)
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    from oracle import prep_ref        # tool: CPU comparison leg only
    from seggroup_amd import prepare, synthetic
    scan = synthetic.make_raw_scan(a.w, a.h, 31, dup_frac=0.02)
    v, f = scan.xyz.shape[0], scan.faces.shape[0]
    mapper = prep_ref.make_mapper(v, a.points, scan.perm)
    dev = "cuda:0"
    d = dict(xyz=torch.from_numpy(scan.xyz).to(dev), rgb=torch.from_numpy(scan.rgb).to(dev), faces=torch.from_numpy(scan.faces).to(dev),
             seg=torch.from_numpy(scan.seg_indices).to(dev), mapper=torch.from_numpy(mapper).to(dev))

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3, r

    out = {"V": v, "F": f, "num_points": a.points}
    ms, (pcl, unmap, missing) = timed(lambda: prepare.sample_points(d["xyz"], d["rgb"], d["mapper"], device=dev))
    out["sample_points_ms"] = round(ms, 3)
    out["unsampled_vertices"] = missing
    out["nearest_pairs_per_s"] = round(missing * a.points / (ms * 1e-3), 0)
    ms, (raw, res) = timed(lambda: prepare.mesh_adjacency(d["faces"], unmap, device=dev))
    out["mesh_adjacency_ms"] = round(ms, 3)
    out["edges_raw"], out["edges_resampled"] = int(raw.shape[0]), int(res.shape[0])
    ms, _ = timed(lambda: prepare.segment_lists(d["seg"], d["mapper"], device=dev))
    out["segment_lists_ms"] = round(ms, 3)
    # CPU legs (oracle, NumPy/torch on the host cores): nearest search on a bounded sample of rows, the rest in full
    miss = np.nonzero(np.bincount(mapper, minlength=v) == 0)[0]
    sample = miss[:2048]
    t = time.perf_counter()
    prep_ref.get_unmapper(scan.xyz[sample], pcl.cpu().numpy()[:, :3])
    dt = time.perf_counter() - t
    out["cpu_oracle"] = {"nearest_pairs_per_s": round(sample.size * a.points / dt, 0), "nearest_sample_rows": int(sample.size)}
    t = time.perf_counter()
    prep_ref.get_adj_from_mesh(scan.faces, unmap.cpu().numpy())
    out["cpu_oracle"]["mesh_adjacency_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    t = time.perf_counter()
    prep_ref.segment_lists(scan.seg_indices, mapper)
    out["cpu_oracle"]["segment_lists_ms"] = round((time.perf_counter() - t) * 1e3, 1)
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
