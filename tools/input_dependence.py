#!/usr/bin/env python3
"""Does the forward 2^24 time depend on the input distribution (power/clock effects)?"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng
n = 1 << 24
P = 0xFFFFFC6E80000001
ntt = eng.NTT(eng.BASELINE_MODULUS, n, enable_inverse=False)
rng = np.random.default_rng(1)
inputs = {
    "randint < 2^62": torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda"),
    "uniform < p": torch.from_numpy(rng.integers(0, P, size=n, dtype=np.uint64).view(np.int64)).cuda(),
    "iota": torch.arange(0x0123456789abcdef, 0x0123456789abcdef + n, dtype=torch.int64, device="cuda"),
    "zeros": torch.zeros(n, dtype=torch.int64, device="cuda"),
}
dst = torch.empty(n, dtype=torch.int64, device="cuda")
K = 300
for rep in range(2):
    for name, src in inputs.items():
        for _ in range(30):
            ntt.compute_forward(dst, src)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            ntt.compute_forward(dst, src)
        torch.cuda.synchronize()
        print("%-16s %.1f us" % (name, (time.perf_counter() - t0) / K * 1e6), flush=True)
