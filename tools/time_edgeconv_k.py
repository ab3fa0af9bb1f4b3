import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import time_edgeconv as T
from seggroup_amd import weights
rng = np.random.default_rng(1)
N = 1200000   # 8 scenes' worth of tiles, one launch
W = weights.make_weights(1, 2.0, affine_jitter=0.3)
x12 = np.zeros((N, 12), np.float32); x12[:, :9] = rng.uniform(-1, 1, (N, 9)).astype(np.float32)
base = (np.arange(N)[:, None] + rng.integers(-300, 300, (N, 40))) % N
for K in (4, 10, 20, 40):
    knn = np.ascontiguousarray(base[:, :K]).astype(np.int32)
    for layers, which in ((1, "mlp_2"), (2, "mlp_3")):
        _, ms = T.run(x12, knn, W, which, layers, reps=5)
        print(f"K={K:3d} layers={layers} total {ms*1e3:8.1f} us")
