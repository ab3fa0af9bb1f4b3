#!/usr/bin/env python3
"""sg_edgeconv_forward alone: accuracy against the oracle on a small case (incl. negative BN gammas) and device time
at scene size.   python tools/time_edgeconv.py [N]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O          # noqa: E402  (tool: checker only)
from seggroup_amd import hip, weights    # noqa: E402

lib = hip.lib()
dev = "cuda:0"


def run(x12, knn, W, which, layers, reps=0):
    N, K = knn.shape
    d_x, d_k = torch.from_numpy(x12).to(dev), torch.from_numpy(knn).to(dev)
    ws = torch.zeros(lib.sg_edgeconv_ws_bytes(N), dtype=torch.uint8, device=dev)
    out = torch.zeros(N, 64, device=dev)
    g = lambda k: torch.from_numpy(W[k]).to(dev)
    keep = [g(f"{which}.conv1.0.weight"), g(f"{which}.bn1.weight"), g(f"{which}.bn1.bias")]
    p2 = (None, None, None)
    if layers == 2:
        keep += [g("mlp_3.conv2.0.weight"), g("mlp_3.bn2.weight"), g("mlp_3.bn2.bias")]
        p2 = tuple(t.data_ptr() for t in keep[3:])
    call = lambda: hip.check(lib.sg_edgeconv_forward(d_x.data_ptr(), d_k.data_ptr(), N, K, layers, *(t.data_ptr() for t in keep[:3]), *p2,
                                                     out.data_ptr(), ws.data_ptr(), ws.numel(), None))
    call()
    torch.cuda.synchronize()
    ms = None
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    return out.cpu().numpy(), ms


def main():
    rng = np.random.default_rng(0)
    # accuracy: random gammas of both signs, one exactly zero
    N, K = 4000, 20
    x9 = rng.uniform(-1, 1, (N, 9)).astype(np.float32)
    x9[:, :3] *= 4
    knn = rng.integers(0, N, (N, K)).astype(np.int32)
    W = weights.make_weights(1, 2.0, affine_jitter=0.3)
    for k in ("mlp_2.bn1.weight", "mlp_3.bn1.weight", "mlp_3.bn2.weight"):
        W[k] = (W[k] * np.where(rng.uniform(size=64) < 0.4, -1.0, 1.0)).astype(np.float32)
        W[k][5] = 0.0
    x12 = np.zeros((N, 12), np.float32)
    x12[:, :9] = x9
    for layers, which in ((1, "mlp_2"), (2, "mlp_3")):
        out, _ = run(x12, knn, W, which, layers)
        ref = O.edgeconv_forward(x9, knn.astype(np.int64), W, which)
        err = np.abs(out - ref)
        print(f"{which}: max err {err.max():.3e} mean {err.mean():.3e} entries > 1e-4: {int((err > 1e-4).sum())}")
    # timing at scene size: neighbours drawn from a +-2000 window (the locality a Morton-ordered kNN table has)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    x12 = np.zeros((N, 12), np.float32)
    x12[:, :9] = rng.uniform(-1, 1, (N, 9)).astype(np.float32)
    knn = ((np.arange(N)[:, None] + rng.integers(-2000, 2000, (N, K))) % N).astype(np.int32)
    W = weights.make_weights(1, 2.0)
    for layers, which, gf in ((1, "mlp_2", 2 * 20 * N * 18 * 64 / 1e9), (2, "mlp_3", 2 * 20 * N * (18 * 64 + 64 * 64) / 1e9)):
        _, ms = run(x12, knn, W, which, layers, reps=20)
        print(f"{which} N={N}: {ms:.4f} ms per forward  ({gf:.1f} GFLOP single evaluation -> {gf / ms:.1f} TFLOP/s algorithmic)")


if __name__ == "__main__":
    main()
