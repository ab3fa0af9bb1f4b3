import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seggroup_amd import hip, weights
from oracle import cpu_ref as O
lib = hip.lib()
rng = np.random.default_rng(0)
N, K = 4000, 20
x9 = rng.uniform(-1, 1, (N, 9)).astype(np.float32); x9[:, :3] *= 4
knn = rng.integers(0, N, (N, K)).astype(np.int32)
W = weights.make_weights(1, 2.0, affine_jitter=0.3)
x12 = np.zeros((N, 12), np.float32); x12[:, :9] = x9
dev = 'cuda:0'
d_x = torch.from_numpy(x12).to(dev); d_k = torch.from_numpy(knn).to(dev)
for layers, which in ((1, 'mlp_2'), (2, 'mlp_3')):
    ws = torch.zeros(lib.sg_edgeconv_ws_bytes(N), dtype=torch.uint8, device=dev)
    out = torch.zeros(N, 64, device=dev)
    g = lambda k: torch.from_numpy(W[k]).to(dev)
    w1, g1, b1 = g(f'{which}.conv1.0.weight'), g(f'{which}.bn1.weight'), g(f'{which}.bn1.bias')
    if layers == 2:
        w2, g2, b2 = g('mlp_3.conv2.0.weight'), g('mlp_3.bn2.weight'), g('mlp_3.bn2.bias')
        p2 = (w2.data_ptr(), g2.data_ptr(), b2.data_ptr())
    else:
        p2 = (None, None, None)
    hip.check(lib.sg_edgeconv_forward(d_x.data_ptr(), d_k.data_ptr(), N, K, layers, w1.data_ptr(), g1.data_ptr(), b1.data_ptr(), *p2,
                                      out.data_ptr(), ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    ref = O.edgeconv_forward(x9, knn.astype(np.int64), W, which)
    err = np.abs(out.cpu().numpy() - ref)
    print(which, 'max err', err.max(), 'mean', err.mean(), 'bad entries', int((err > 1e-4).sum()))
    print('  per-channel max err', np.round(err.max(0), 6))
    bad_rows = np.nonzero((err > 1e-4).any(1))[0]
    print('  bad rows', bad_rows[:20], len(bad_rows))
