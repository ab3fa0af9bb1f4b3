#!/usr/bin/env python3
"""Work counters of the in-cluster kNN kernels (profiling build only: make -C seggroup_amd/csrc PROFILE=1, SG_KNN_DEBUG=32):
tiles, chunk box tests, chunks scanned, segment tests, keys appended, drain steps -- per scene, for the unseeded pair of layers
(variant 1) and the default (layer 3 seeded)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SG_KNN_DEBUG", sys.argv[2] if len(sys.argv) > 2 else "48")
import torch  # noqa: E402
from seggroup_amd import hip, synthetic, weights  # noqa: E402
from seggroup_amd.model import Pipeline  # noqa: E402
from seggroup_amd.scene import DeviceScene  # noqa: E402
lib = hip.lib()
if not hasattr(lib, "sg_debug_knn5_stats"):
    raise SystemExit("release build: make -C seggroup_amd/csrc clean && make -C seggroup_amd/csrc PROFILE=1")
W = weights.load_npz(os.path.join(ROOT, "tests/golden/weights_g2.npz"))
prof = sys.argv[1] if len(sys.argv) > 1 else "voronoi"
sc = synthetic.make_scene(150000, 1500, 30000 if prof == "voronoi" else 70100, **({} if prof == "voronoi" else {"seg_profile": prof}))
print("profile", prof, "largest segment", int(__import__("numpy").bincount(sc.seg).max()))
ds = DeviceScene.from_synthetic(sc, "cuda:0")
pl = Pipeline(W, ds.N, ds.S, ds.E0, ds.V, device="cuda:0")
buf = (C.c_ulonglong * 16)()
for variant, what in ((1, "both layers unseeded"), (8, "layer 2 unseeded + layer 3 seeded")):
    pl.lib.sg_pipeline_set_knn_variant(pl.handle, variant)
    pl.forward(ds, hip.MODE_INS_INFER)
    lib.sg_debug_knn5_stats(buf)
    res = pl.forward(ds, hip.MODE_INS_INFER)
    torch.cuda.synchronize()
    lib.sg_debug_knn5_stats(buf)
    v = list(buf)
    t = max(v[0], 1)
    print(f"{what}: tiles {v[0]} | per tile: chunk tests {v[6] / t:.1f}, chunks scanned {v[5] / t:.1f} (= {32 * v[5] / t:.0f} candidates), segment tests {v[7] / t:.1f}, "
          f"slowest tile {v[13] >> 40} cycles (its cluster: {((v[13] >> 26) & 0x3fff) * 64} points in {(v[13] >> 14) & 0xfff} segments; {v[13] & 0x3fff} chunks scanned), {v[15]} tiles in clusters > 2048 points averaging {v[14] / max(v[15], 1):.0f} cycles (all tiles: {(v[1] + v[2] + v[3] + v[4]) / t:.0f}), "
          f"keys appended per lane {v[8] / t / 64:.1f}, drain steps {v[9] / t:.1f} in {v[10] / t:.1f} drains ({v[11] / t:.1f} empty, {v[12] / t:.1f} merged by twelve) | cycles per tile: phase A {v[1] / t:.0f}, B {v[3] / t:.0f}, out {v[4] / t:.0f}; trace {res.trace}")
