#!/usr/bin/env python3
"""Adds the ORACLE's label digests for the digest-only fixtures (150k / 500k) to tests/golden/index.json, next to
the reference's.  The GPU test at those sizes checks HIP == oracle digests (same defined tie rule) and HIP ==
reference digests (holds on reference-stable seeds)."""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seggroup_amd import synthetic, weights
from oracle import cpu_ref
idx_path = os.path.join(ROOT, "tests", "golden", "index.json")
idx = json.load(open(idx_path))
W = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g2.npz"))
for name in sys.argv[1:] or ["scene_150k", "stress_500k"]:
    e = idx[name]
    sc = synthetic.make_scene(e["n"], e["s"], e["seed"], **e["kw"])
    t = time.time()
    r = cpu_ref.forward_scene(sc, W, "ins_infer")
    e["ins_infer"]["oracle_label_sha"] = {k: hashlib.sha256(np.ascontiguousarray(v.astype(np.int32)).tobytes()).hexdigest()
                                          for k, v in r["labels"].items()}
    e["ins_infer"]["oracle_trace"] = r["trace"]
    e["ins_infer"]["oracle_equals_reference"] = e["ins_infer"]["oracle_label_sha"] == e["ins_infer"]["label_sha"]
    print(name, "oracle", round(time.time() - t, 1), "s  trace", r["trace"], " == reference digests:", e["ins_infer"]["oracle_equals_reference"], flush=True)
    if "sem_infer" in e:                                           # fixtures captured with sem=True (weights_g1, th = 3: returns after the structural layer)
        W1 = weights.load_npz(os.path.join(ROOT, "tests", "golden", "weights_g1.npz"))
        r = cpu_ref.forward_scene(sc, W1, "sem_infer")
        e["sem_infer"]["oracle_label_sha"] = {k: hashlib.sha256(np.ascontiguousarray(v.astype(np.int32)).tobytes()).hexdigest()
                                              for k, v in r["labels"].items()}
        e["sem_infer"]["oracle_trace"] = r["trace"]
        e["sem_infer"]["oracle_equals_reference"] = e["sem_infer"]["oracle_label_sha"] == e["sem_infer"]["label_sha"]
        print(name, "sem_infer oracle trace", r["trace"], " == reference digests:", e["sem_infer"]["oracle_equals_reference"], flush=True)
    json.dump(idx, open(idx_path, "w"), indent=1, sort_keys=True)
