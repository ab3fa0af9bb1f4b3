#!/usr/bin/env python3
"""One scene at a time through sg_pipeline_forward (the single-scene pipeline on the default stream): wall time per forward and the
pipeline's own stage times (HIP events).  The program tools/stress_500k.sh profiles for BASELINE.json configs[4].

    python tools/time_scene.py N S [iterations]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seggroup_amd import hip, synthetic, weights  # noqa: E402
from seggroup_amd.model import SegModel  # noqa: E402
from seggroup_amd.scene import DeviceScene  # noqa: E402

n, s = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
W = weights.load_npz(os.path.join(ROOT, "tests/golden/weights_g2.npz"))
t = time.time()
sc = synthetic.make_scene(n, s, 50005 if n == 500000 else 20004)
print("scene generated in %.1f s: N %d S %d E0 %d V %d" % (time.time() - t, sc.num_points, sc.num_segments, sc.adj.shape[0], sc.unmap.shape[0]))
net = SegModel(exp_name="t", ins_infer=True)
net.load_weights(W)
net.epoch = "ins_infer"
ds = DeviceScene.from_synthetic(sc, "cuda:0")
pipe = net.pipeline_for(ds)
print("pipeline device MB %.1f" % (pipe.device_bytes() / 1e6))
if os.environ.get("SG_KNN_VARIANT"):                      # 1 / 2 / 4 waves per tile of the one-pass kernel, 0 = two-pass, 8 = one wave + seeded layer 3
    hip.lib().sg_pipeline_set_knn_variant(pipe.handle, int(os.environ["SG_KNN_VARIANT"]))
walls, st = [], {}
for it in range(iters):
    torch.cuda.synchronize()
    t = time.time()
    res = pipe.forward(ds, hip.MODE_INS_INFER)
    walls.append((time.time() - t) * 1e3)
    st = pipe.stage_times()
    print(f"iter {it}: wall {walls[-1]:.2f} ms  gpu-stage-sum {sum(v for k, v in st.items() if k.count('.') <= 1):.2f} ms trace {res.trace} fallback {res.used_fallback}")
print(json.dumps({"points": n, "segments": s, "E0": int(sc.adj.shape[0]), "V": int(sc.unmap.shape[0]), "wall_ms_median": round(float(np.median(walls[2:])), 3),
                  "wall_ms_min": round(min(walls[2:]), 3), "trace": list(res.trace), "stage_ms": {k: round(v, 4) for k, v in st.items() if v > 0}}))
if os.environ.get("SG_HOST_PROFILE"):
    # where the host's share of the wall time goes (pipeline.cpp prints the split to stderr at the end of an sg_batch_forward call)
    import ctypes as C
    cnt = 16
    pipes = (C.c_void_p * 1)(pipe.handle)
    c_scenes = (hip.Scene * cnt)(*[ds.c_struct for _ in range(cnt)])
    c_res = (hip.Result * cnt)()
    for i in range(cnt):
        c_res[i].h_labels = pipe.labels.data_ptr()
    hip.check(hip.lib().sg_batch_forward(pipes, 1, c_scenes, cnt, hip.MODE_INS_INFER, c_res, None, None, None, 0))
