#!/usr/bin/env python3
"""How much do per-pass HIP events cost bench.py's timed region?  Same loop with events
around every pass, with events on every 8th step, and with none."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng
n = 1 << 24
ntt = eng.NTT(eng.BASELINE_MODULUS, n, enable_inverse=False)
src = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
dst = torch.empty_like(src)
npass = ntt.num_passes()
K = 200
def run(every):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(npass + 1)] for _ in range(K)]
    for _ in range(20):
        ntt.compute_forward(dst, src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        e = ev[k] if (every and k % every == 0) else None
        for i in range(npass):
            if e: e[i].record()
            ntt.run_pass(False, i, dst, src if i == 0 else None)
        if e: e[npass].record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for rep in range(3):
    print("events every step: %.1f us | every 8th: %.1f us | none: %.1f us | compute_forward: " % (run(1), run(8), run(0)), end="")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): ntt.compute_forward(dst, src)
    torch.cuda.synchronize(); print("%.1f us" % ((time.perf_counter() - t0) / K * 1e6))
