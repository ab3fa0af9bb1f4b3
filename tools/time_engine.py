#!/usr/bin/env python3
"""Solo-batched stage times of the scene engine (one group of 8 scenes alone on the GPU, HIP events on its stream): a quick
before / after for kernel work, e.g. under different development knobs (environment variables read by the library).

    python tools/time_engine.py [--scenes 16] [--groups 1] [--per-group 8] [--rounds 4] [--scene-cache DIR] [--profile voronoi]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--per-group", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--scene-cache", default=os.environ.get("SG_SCENE_CACHE", ""))
    ap.add_argument("--profile", default="voronoi")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    jobs = [(150000, 1500, (30000 if a.profile == "voronoi" else 70100) + i, a.profile, a.scene_cache) for i in range(a.scenes)]
    it, pool = bench.generate_scenes(jobs, 1 if a.scene_cache and all(os.path.exists(os.path.join(a.scene_cache, f"scene_{j[3]}_{j[0]}_{j[1]}_{j[2]}.npz")) for j in jobs) else 16)
    import torch  # noqa: E402
    from seggroup_amd import hip, weights  # noqa: E402
    from seggroup_amd.model import Engine  # noqa: E402
    from seggroup_amd.scene import DeviceScene  # noqa: E402
    scenes = [DeviceScene.from_synthetic(s, device="cuda:0") for s in it]
    if pool is not None:
        pool.shutdown()
    W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    eng = Engine(W, caps, groups=a.groups, per_group=a.per_group, device="cuda:0", timing=1)
    eng.run(scenes, hip.MODE_INS_INFER)
    eng.reset_stage_stats()
    import time  # noqa: E402
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.rounds):
        eng.run(scenes, hip.MODE_INS_INFER)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms = eng.mean_stage_ms()
    keys = ["kernel.l2.edgeconv", "evaluate", "l2.gcn+dist", "l3.gcn+dist", "dist1+d2h", "kernel.l3.edgeconv", "l2.knn", "l3.knn", "l3.edgeconv.stats1", "mlp1", "fps64", "l2.gather", "l3.gather", "contract_edges"]
    print(json.dumps({"tag": a.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("SG_")}, "scenes_per_s": round(a.rounds * len(scenes) / dt, 1),
                      "us_per_scene": {k: round(ms.get(k, 0.0) * 1e3, 1) for k in keys},
                      "sum_us_per_scene": round(sum(v for k, v in ms.items() if k.count(".") <= 1) * 1e3, 1)}))
    eng.close()


if __name__ == "__main__":          # the scene generator's pool re-imports this module in its workers
    main()
