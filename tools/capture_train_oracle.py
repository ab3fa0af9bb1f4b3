#!/usr/bin/env python3
"""Full-size gradient vectors of the training step from the float64 oracle chain (oracle/train_ref.py; itself pinned to the real
reference's gradients on the small fixtures), for workloads of tests/golden/seed_scan.json whose forward structure is known to
agree between the engine, the oracle and the reference.  The autograd graph of a 150k-point scene holds ~20 GB of float64
activations, so this runs in the build container once and the GPU test compares against the stored vectors:

    tests/golden/train_grads_full.npz   <workload>.<seed>.grad  float32 [147880] in named_parameters() order
                                        <workload>.<seed>.loss  [loss_sum, K]

usage: python tools/capture_train_oracle.py [workload:seed ...]      (default: uniform_150k:20000 scannet_150k:70010)
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("SEGGROUP_HOST_ONLY", "1")


def main():
    from oracle import train_ref
    from seggroup_amd import synthetic, trainer as T, weights as W
    book = json.load(open(os.path.join(REPO, "tests", "golden", "seed_scan.json")))
    wts = dict(W.load_npz(os.path.join(REPO, "tests", "golden", "weights_g2.npz")))
    gt = np.load(os.path.join(REPO, "tests", "golden", "train_tail.npz"))
    wts.update({k[2:]: gt[k] for k in gt.files if k.startswith("w.")})
    path = os.path.join(REPO, "tests", "golden", "train_grads_full.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for item in sys.argv[1:] or ["uniform_150k:20000", "scannet_150k:70010"]:
        name, seed = item.split(":")
        e = book[name]
        scene = synthetic.make_scene(e["n"], e["s"], int(seed), name=f"scene{int(seed):05d}_00", **e["kw"])
        t0 = time.time()
        r = train_ref.training_step(scene, wts)
        flat = np.zeros(T.NUM_PARAMS, np.float32)
        for pname, off, cnt in T.param_slots():
            flat[off:off + cnt] = r["grads"][pname].reshape(-1)
        out[f"{name}.{seed}.grad"] = flat
        out[f"{name}.{seed}.loss"] = r["loss"]
        print(item, "loss", r["loss"], "%.0f s" % (time.time() - t0), flush=True)
        np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
