import sys, os, json, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seggroup_amd import weights, synthetic, hip
from seggroup_amd.model import SegModel
from seggroup_amd.scene import DeviceScene
n, s = int(sys.argv[1]), int(sys.argv[2])
W = weights.load_npz(os.path.join(ROOT, 'tests/golden/weights_g2.npz'))
t = time.time(); sc = synthetic.make_scene(n, s, 20004); print('gen', time.time() - t)
net = SegModel(exp_name='t', ins_infer=True); net.load_weights(W); net.epoch = 'ins_infer'
ds = DeviceScene.from_synthetic(sc, 'cuda:0')
pipe = net.pipeline_for(ds)
print('pipeline device MB', pipe.device_bytes() / 1e6)
for it in range(4):
    torch.cuda.synchronize(); t = time.time()
    res = pipe.forward(ds, hip.MODE_INS_INFER)
    dt = time.time() - t
    st = pipe.stage_times()
    print(f'iter {it}: wall {dt*1e3:.2f} ms  gpu-stage-sum {sum(st.values()):.2f} ms trace {res.trace} fallback {res.used_fallback}')
print({k: round(v, 3) for k, v in st.items()})
import ctypes
lib = hip.lib()
if not hasattr(lib, "sg_debug_knn_stats"):
    print("(release build: the kNN work counters are compiled out -- make -C seggroup_amd/csrc PROFILE=1 keeps them)")
    sys.exit(0)

buf = (ctypes.c_ulonglong * 8)()
lib.sg_debug_knn_stats(buf, 1)
res = pipe.forward(ds, hip.MODE_INS_INFER)
lib.sg_debug_knn_stats(buf, 1)
print('raw', list(buf)); print('knn stats (one scene, both layers): scanned(wave-cands) %d  appends(lane) %d  drain-iters(wave) %d  segs visited %d skipped %d' % tuple(buf[:5]))

import numpy as np
bt = (ctypes.c_ulonglong * 8192)()
lib.sg_debug_knn_blocktimes(bt, 8192)
res = pipe.forward(ds, hip.MODE_INS_INFER)
lib.sg_debug_knn_blocktimes(bt, 8192)      # holds the LAST knn launch (layer 3) of that forward
a = np.array(bt[:], dtype=np.float64); a = a[a > 0]
if a.size:
    print('layer-3 knn wave runtimes (shader clocks): waves %d  sum %.3e  mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f' % (
        a.size, a.sum(), a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99), a.max()))

b5 = (ctypes.c_ulonglong * 16)()
lib.sg_debug_knn5_stats(b5)
res = pipe.forward(ds, hip.MODE_INS_INFER)
lib.sg_debug_knn5_stats(b5)
v = list(b5)
if v[0]:
    nb = v[0]
    print('knn5 (one scene): blocks %d | per-wave cycles: phaseA %.0f mergeA %.0f phaseB %.0f final %.0f | per block: chunks scanned %.1f tested %.1f segs tested %.1f appends/lane %.1f drain iters/wave %.1f' % (
        nb, v[1] / nb / 4, v[2] / nb / 4, v[3] / nb / 4, v[4] / nb / 4, v[5] / nb, v[6] / nb, v[7] / nb, v[8] / nb / 64, v[9] / nb / 4))
