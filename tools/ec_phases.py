#!/usr/bin/env python3
"""Profiling build only (make -C seggroup_amd/csrc clean all PROFILE=1): where the EdgeConv workgroups spend their time.
Runs the engine on 8 scenes (one group) a few times and prints the per-phase share of wave time per kernel."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from seggroup_amd import hip, model, synthetic, weights
from seggroup_amd.scene import DeviceScene
lib = hip.lib()
if not hasattr(lib, "sg_debug_ec_phases"):
    sys.exit("not a profiling build")
w = weights.make_weights(1, 2.0)
scenes = [DeviceScene.from_synthetic(synthetic.make_scene(150000, 1500, seed=20000 + i), "cuda:0") for i in range(8)]
eng = model.BatchRunner(w, scenes, inflight=8, per_group=8)
buf = (C.c_ulonglong * 24)()
for rep in range(3):
    eng.run(scenes)
    lib.sg_debug_ec_phases(buf)
names = ["stage weights", "own row + base", "slot loop", "statistics flush", "maxima out", "barrier + partials"]
for m, nm in ((0, "compiler loop"), (1, "MLP2 hand"), (2, "MLP3 hand")):
    row = [buf[m * 8 + k] for k in range(6)]
    tot = sum(row)
    if tot == 0:
        continue
    print(nm, "total wave time %.1f ms" % (tot * 1e-5), " | ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, row)))
