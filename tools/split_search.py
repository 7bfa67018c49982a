#!/usr/bin/env python3
"""For batched transforms of 2^24 elements in total, time every admissible column length n0
(wall clock, stream of launches) next to the planner's own choice (n0 = 0)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sve_ntt_amd as eng

total_log2 = 24
for log2n in range(14, 22):
    n, batch = 1 << log2n, 1 << (total_log2 - log2n)
    x = torch.randint(0, 1 << 62, (n * batch,), dtype=torch.int64, device="cuda")
    y = torch.empty_like(x)
    res = []
    for n0 in [0] + list(range(1, min(12, log2n - 1) + 1)):
        try:
            ntt = eng.NTT(eng.BASELINE_MODULUS, n, n0_log2=n0, batch=batch, enable_inverse=False)
        except ValueError:
            continue
        if ntt.num_passes() > 2:
            continue
        for _ in range(200):
            ntt.compute_forward(y, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            ntt.compute_forward(y, x)
        torch.cuda.synchronize()
        res.append((n0, (time.perf_counter() - t0) / 100 * 1e6, ntt.describe()))
    best = min(res, key=lambda r: r[1])
    print(f"n=2^{log2n} x {batch}: planner {res[0][1]:.1f} us [{res[0][2]}]; best n0={best[0]} {best[1]:.1f} us [{best[2]}]")
    print("    " + "  ".join(f"{n0}:{t:.0f}" for n0, t, _ in res[1:]), flush=True)
