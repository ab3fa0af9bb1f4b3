#!/usr/bin/env python3
"""Wall time of one training step on the GPU box: forward (with tape) / loss / backward / optimizer on synthetic
150k-point scenes, N distinct scenes cycled.   usage: python tools/time_train.py [--points 150000 --segments 1500 --steps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--segments", type=int, default=1500)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--profile", default="uniform")
    ap.add_argument("--lanes", type=int, default=1, help="> 1: BatchTrainer with this many scenes per optimizer step (each on its own stream)")
    a = ap.parse_args()
    import torch
    from seggroup_amd import synthetic, train, trainer as T, weights as W
    from seggroup_amd.scene import DeviceScene
    wts = W.to_state_dict(W.load_npz(os.path.join(REPO, "tests", "golden", "weights_g2.npz")), prefix="")
    state = train.initial_state(1)
    state.update({k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in wts.items()})
    kw = dict(seg_profile="scannet") if a.profile == "scannet" else {}
    scenes = [DeviceScene.from_synthetic(synthetic.make_scene(a.points, a.segments, 20000 + i, name=f"scene{i:04d}_00", **kw), device="cuda:0")
              for i in range(a.scenes)]
    caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
    if a.lanes > 1:
        a.scenes = max(a.scenes, a.lanes)
        while len(scenes) < a.scenes:
            i = len(scenes)
            scenes.append(DeviceScene.from_synthetic(synthetic.make_scene(a.points, a.segments, 20000 + i, name=f"scene{i:04d}_00", **kw), device="cuda:0"))
        caps = (max(s.N for s in scenes), max(s.S for s in scenes), max(s.E0 for s in scenes), max(s.V for s in scenes))
        bt = T.BatchTrainer(state, caps, lanes=a.lanes, device="cuda:0")
        losses, t_all = [], 0.0
        for step in range(-2, a.steps):
            group = [scenes[(step * a.lanes + k) % len(scenes)] for k in range(a.lanes)]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ls, _, _ = bt.step(group)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if step >= 0:
                t_all += t1 - t0
                losses.append(float(np.mean([l[0, 0] / l[0, 1] for l in ls])))
        print(json.dumps({"lanes": a.lanes, "step_ms": round(t_all / a.steps * 1e3, 3), "ms_per_scene": round(t_all / a.steps / a.lanes * 1e3, 3),
                          "points": a.points, "segments": a.segments, "steps": a.steps, "first_loss": round(losses[0], 4), "last_loss": round(losses[-1], 4),
                          "device_mb": round(sum(l.device_bytes() for l in bt.lanes) / 2 ** 20, 1)}))
        bt.close()
        return
    tr = T.Trainer(state, caps, device="cuda:0")
    t = dict(forward=0.0, loss=0.0, backward=0.0, optimizer=0.0)
    losses = []
    for step in range(-2, a.steps):
        sc = scenes[step % len(scenes)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.forward(sc)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        mask = tr.dropout_mask("random")
        loss = tr.loss(mask)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tr.backward(mask)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        tr.average_gradients()
        tr.optimizer_step()
        tr.update_running_stats()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        if step >= 0:
            t["forward"] += t1 - t0; t["loss"] += t2 - t1; t["backward"] += t3 - t2; t["optimizer"] += t4 - t3
            losses.append(float(loss[0, 0] / loss[0, 1]))
    out = {k: round(v / a.steps * 1e3, 3) for k, v in t.items()}
    out["step_ms"] = round(sum(t.values()) / a.steps * 1e3, 3)
    out.update(points=a.points, segments=a.segments, steps=a.steps, profile=a.profile, first_loss=round(losses[0], 4), last_loss=round(losses[-1], 4),
               device_mb=round(tr.device_bytes() / 2 ** 20, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
