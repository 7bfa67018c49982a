#!/usr/bin/env python3
"""us per forward 2^24 transform in consecutive chunks of 100 launches from a cold start."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng
n = 1 << 24
ntt = eng.NTT(eng.BASELINE_MODULUS, n, enable_inverse=False)
src = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
dst = torch.empty_like(src)
torch.cuda.synchronize()
time.sleep(2.0)  # let the GPU fall idle
out = []
for chunk in range(30):
    t0 = time.perf_counter()
    for _ in range(100):
        ntt.compute_forward(dst, src)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 100 * 1e6)
print(" ".join("%.0f" % x for x in out))
