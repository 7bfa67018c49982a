#!/usr/bin/env python3
"""Batched single-pass transforms, 2^24 elements in all: us per launch for n = 2^1..2^13."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng
tot = 24
x = torch.randint(0, 1 << 62, (1 << tot,), dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
for inverse in (False, True):
    for log2n in range(1, 14):
        ntt = eng.NTT(eng.BASELINE_MODULUS, 1 << log2n, batch=1 << (tot - log2n))
        fn = ntt.compute_inverse if inverse else ntt.compute_forward
        for _ in range(300):
            fn(y, x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            fn(y, x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        print("%s n=2^%-2d %-26s %7.1f us  %6.0f GB/s" % ("inv" if inverse else "fwd", log2n, ntt.describe(), dt * 1e6, 16 * (1 << tot) / dt / 1e9), flush=True)
