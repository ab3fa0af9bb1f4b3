#!/usr/bin/env python3
"""Times the EdgeConv operator alone (sg_edgeconv_forward_x) with the hand-scheduled and the compiler-scheduled slot loop.
   python tools/time_edgeconv.py [N] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from seggroup_amd import hip, weights

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lib = hip.lib()
    rng = np.random.default_rng(5)
    K = 20
    x12 = np.zeros((N, 12), np.float32)
    x12[:, :9] = rng.uniform(-1, 1, (N, 9)).astype(np.float32)
    # clusters of ~150 consecutive rows: neighbours come from the row's own block (what member order gives the real layers)
    blk = np.arange(N) // 150
    knn = (blk[:, None] * 150 + rng.integers(0, 150, (N, K))).clip(0, N - 1).astype(np.int32)
    W = weights.make_weights(1, 2.0, affine_jitter=0.3)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_x, d_k = up(x12), up(knn)
    t = {k: up(W[k]) for k in W}
    ws = torch.empty(lib.sg_edgeconv_ws_bytes(N), dtype=torch.uint8, device="cuda")
    out = torch.zeros(N, 64, device="cuda")
    res = {}
    for layers, which in ((1, "mlp_2"), (2, "mlp_3")):
        p2 = (t["mlp_3.conv2.0.weight"].data_ptr(), t["mlp_3.bn2.weight"].data_ptr(), t["mlp_3.bn2.bias"].data_ptr()) if layers == 2 else (None, None, None)
        for flags in (0, 1):
            def call():
                rb = torch.zeros(256, dtype=torch.int32, device="cuda")
                hip.check(lib.sg_edge_range(d_x.data_ptr(), N, rb.data_ptr(), None))
                hip.check(lib.sg_edgeconv_forward_x(d_x.data_ptr(), d_k.data_ptr(), N, K, layers, t[f"{which}.conv1.0.weight"].data_ptr(),
                                                    t[f"{which}.bn1.weight"].data_ptr(), t[f"{which}.bn1.bias"].data_ptr(), *p2, out.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), rb.data_ptr(), flags, None))
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            res[(which, flags)] = ms
            print(f"{which} N={N} {'compiler loop' if flags else 'hand-scheduled'}: {ms:.3f} ms per call (whole operator: range + moments/fold + edgeconv + apply)", flush=True)
            h = out.cpu().numpy()
            print("   checksum", float(np.abs(h).sum()), "finite", bool(np.isfinite(h).all()))
main()
