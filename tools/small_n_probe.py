#!/usr/bin/env python3
"""Small-transform latency probe: runs N=2^17 (as 2^8 x 2^9), 2^10 and 2^19 forward 200x
each; meant to be run under `rocprofv3 --kernel-trace --stats` (kernel durations) and also
prints the wall time per transform for a back-to-back stream of launches."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng

for log2n, n0 in ((17, 8), (10, 0), (19, 0), (14, 0)):
    n = 1 << log2n
    ntt = eng.NTT(eng.BASELINE_MODULUS, n, n0_log2=n0)
    x = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
    y = torch.empty_like(x)
    for _ in range(20):
        ntt.compute_forward(y, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ntt.compute_forward(y, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"2^{log2n}: {ntt.describe()}: {dt*1e6:.1f} us per transform (launch stream, wall)")
