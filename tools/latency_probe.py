#!/usr/bin/env python3
"""Wall time of one scene alone on the GPU (sg_pipeline_forward, stage timing off), the way bench.py's extra.latency_ms_single_scene measures it.

    python3 tools/latency_probe.py [N=150000] [S=1500] [iterations=40]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from seggroup_amd import hip, synthetic, weights
from seggroup_amd.model import Pipeline
from seggroup_amd.scene import DeviceScene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
s = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
sc = DeviceScene.from_synthetic(synthetic.make_scene(n, s, 50005 if n == 500000 else 20000), device="cuda:0")
pl = Pipeline(W, sc.N, sc.S, sc.E0, sc.V, stream=None, device="cuda:0")
pl.set_timing(0)
ts = []
for it in range(iters):
    torch.cuda.synchronize(); t = time.perf_counter()
    pl.forward(sc, hip.MODE_INS_INFER)
    ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts[5:])
print(f"N {n} S {s}: median {np.median(ts):.3f} ms, min {ts.min():.3f}, p90 {np.percentile(ts, 90):.3f}  (SG_SYNC_SPIN={'1' if os.environ.get('SG_SYNC_SPIN') else '0'})")
