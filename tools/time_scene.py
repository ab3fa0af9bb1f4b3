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
