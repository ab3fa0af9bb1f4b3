#!/usr/bin/env python3
"""before the CPU port's number on the GPU box is trusted, its faithful mode must land within ~+-30 % of the REAL
reference's 24.2 s per 150k-point scene in the BUILD container (8 vCPU).  Runs oracle/cpu_ref.py there on the bench's scene 0 and
writes profiles/<tag>_cpu_anchor_check.json (bench.py copies it into cpu_baseline.anchor_check).

    python tools/cpu_anchor_check.py r04
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
from oracle import cpu_ref  # noqa: E402
from seggroup_amd import synthetic, weights  # noqa: E402

W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
sc = synthetic.make_scene(150000, 1500, seed=20000)          # bench.py's scene 0 (configs[1] / configs[2])
cores = os.cpu_count()
torch.set_num_threads(cores)
times = []
for _ in range(2):
    t = time.perf_counter()
    ref = cpu_ref.forward_scene(sc, W, "ins_infer", faithful=True)
    times.append(time.perf_counter() - t)
best = min(times)
out = {"anchor_s_per_scene_real_reference": 24.2, "anchor_source": "BASELINE.md section 2: the unmodified reference, this container, 8 vCPU",
       "port_s_per_scene_faithful": [round(x, 2) for x in times], "port_best_s": round(best, 2), "ratio_port_over_reference": round(best / 24.2, 3),
       "within_30_percent": bool(abs(best / 24.2 - 1.0) <= 0.30), "cores": cores, "trace": list(map(int, ref["trace"])),
       "what": "oracle/cpu_ref.py faithful mode (Python-loop edge contraction and text formatting like the reference; kNN / export vectorised) on "
               "bench.py's scene 0, build container"}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", f"{tag}_cpu_anchor_check.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
