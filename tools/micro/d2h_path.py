#!/usr/bin/env python3
"""Which engine moves a device -> pinned-host copy of a label block (8.4 MB) on a non-blocking stream behind a kernel: the copy engines (SDMA) or a
blit KERNEL on the CUs (`__amd_rocclr_copyBuffer` in a kernel trace)?  Run under `rocprofv3 --kernel-trace --stats` with different runtime knobs.

    python3 tools/micro/d2h_path.py [bytes] [copies]
"""
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_400_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
src = torch.zeros(n, dtype=torch.uint8, device=dev)
dst = torch.empty(n, dtype=torch.uint8, pin_memory=True)
a = torch.zeros(1 << 20, device=dev)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    for _ in range(5):
        a.add_(1.0)
        dst.copy_(src, non_blocking=True)
    st.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        a.add_(1.0)
        dst.copy_(src, non_blocking=True)
    st.synchronize()
    dt = time.perf_counter() - t
print("%d copies of %.1f MB behind a kernel each: %.1f us per (kernel + copy), %.1f GB/s" % (reps, n / 1e6, dt / reps * 1e6, n * reps / dt / 1e9))
