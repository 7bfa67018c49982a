#!/usr/bin/env python3
"""Forward-transform time across lengths (stream of launches, wall clock, inputs resident):
    python tools/sweep.py [lo] [hi] > gpurun_out/sweep.txt"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 28
print("%-6s %-58s %10s %12s %8s" % ("n", "plan", "us", "elem/s", "GB/s"))
for log2n in range(lo, hi + 1):
    n = 1 << log2n
    ntt = eng.NTT(eng.BASELINE_MODULUS, n, device_pointers=True)
    x = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda")
    y = torch.empty_like(x)
    reps = max(50, min(2000, (1 << 31) // n))  # ~25 ms of work at least: steady-state clocks (tools/clock_ramp.py)
    for _ in range(reps):
        ntt.compute_forward(y, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ntt.compute_forward(y, x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("2^%-4d %-58s %10.1f %12.3e %8.0f" % (log2n, ntt.describe(), dt * 1e6, n / dt, 16 * n / dt / 1e9), flush=True)
    del x, y, ntt
